"""Click command groups of the device backend (features / dsp / filter): see main.py."""
