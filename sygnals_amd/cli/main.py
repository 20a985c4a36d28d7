"""Click commands of the device backend.

`features extract` keeps the option surface of sygnals/cli/features_cmd.py:30-113
(-o/--output, repeatable -f/--feature incl. 'all', --frame-length 2048, --hop-length 512;
.csv -> DataFrame, .npz -> dict of arrays; ValueError -> click.UsageError).  The `dsp` and
`filter` groups follow the surface documented in the reference's README.md:483-533, 704-900 --
in the reference snapshot those groups are commented out (sygnals/cli/main.py:29-33, 115-119), so
they are provided here rather than kept.  Run stand-alone as `python -m sygnals_amd.cli.main ...`
or attach the groups to the reference CLI through the plugin (register_cli_commands).
"""
from __future__ import annotations

import logging
from pathlib import Path

import click
import numpy as np
import pandas as pd

from .. import io as sio

logger = logging.getLogger(__name__)


@click.group("sygnals-amd")
def cli():
    """MI355X backend for the sygnals feature-extraction hot path."""


# ---------------------------------------------------------------- features
@click.group("features")
def features_cmd():
    """Extract, transform, and manage signal features."""


@features_cmd.command("extract")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False, resolve_path=True))
@click.option("-o", "--output", type=click.Path(resolve_path=True), required=True,
              help="Output file path for extracted features (e.g., features.csv, features.npz).")
@click.option("-f", "--feature", "features", multiple=True, required=True,
              help="Feature(s) to extract (e.g., 'spectral_centroid', 'mfcc'). Use 'all'. Can be repeated.")
@click.option("--frame-length", type=int, default=2048, show_default=True, help="Analysis frame length (samples).")
@click.option("--hop-length", type=int, default=512, show_default=True, help="Hop length between frames (samples).")
def features_extract(input_file, output, features, frame_length, hop_length):
    """Extract features from an audio signal."""
    from ..core.features.manager import extract_features
    input_path, output_path = Path(input_file), Path(output)
    feature_list = list(features)
    if len(feature_list) == 1 and feature_list[0].lower() == "all":
        feature_list = ["all"]
    try:
        res = sio.read_data(input_path)
        if not isinstance(res, tuple) or len(res) != 2:
            raise click.UsageError(f"Input file '{input_path.name}' is not recognized as audio.")
        signal, sr = res
        if signal.ndim != 1:
            logger.warning("Input audio is multi-channel. Converting to mono by averaging for feature extraction.")
            signal = np.mean(signal, axis=0)
        fmt = "dict_of_arrays" if output_path.suffix.lower() == ".npz" else "dataframe"
        out = extract_features(y=signal, sr=sr, features=feature_list, frame_length=frame_length,
                               hop_length=hop_length, output_format=fmt)
        if (isinstance(out, pd.DataFrame) and out.empty) or (isinstance(out, dict) and
                                                             not any(k != "time" for k in out)):
            click.echo("Warning: No features extracted or signal too short.")
            return
        sio.save_data(out, output_path)
        click.echo(f"Successfully extracted features from '{input_path.name}' and saved to '{output_path.name}'.")
    except FileNotFoundError:
        raise click.UsageError(f"Input file not found: {input_path}")
    except ValueError as e:
        raise click.UsageError(f"Error during feature extraction: {e}")


# ---------------------------------------------------------------- dsp
@click.group("dsp")
def dsp_cmd():
    """Perform core Digital Signal Processing (DSP) operations."""


def _load_signal(path, fs):
    x, sr = sio.signal_from(sio.read_data(path))
    if x.ndim != 1:
        x = np.mean(x, axis=0)
    return x, (fs if fs is not None else sr)


@dsp_cmd.command("fft")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--output", required=True, type=click.Path())
@click.option("--fs", type=float, required=True, help="Sampling frequency (Hz).")
@click.option("--window", default="hann", show_default=True)
@click.option("--n", type=int, default=None, help="FFT length. Defaults to signal length.")
def dsp_fft(input_file, output, fs, window, n):
    """Compute the Fast Fourier Transform (FFT)."""
    from ..core.dsp import compute_fft
    try:
        x, _ = _load_signal(input_file, fs)
        freqs, spec = compute_fft(x, fs=fs, n=n, window=window if window and window.lower() != "none" else None)
    except ValueError as e:
        raise click.UsageError(str(e))
    if Path(output).suffix.lower() == ".npz":
        sio.save_data({"frequencies": freqs, "spectrum": spec, "fs": np.array(fs)}, output)
    else:
        sio.save_data(pd.DataFrame({"Frequency": freqs, "Magnitude": np.abs(spec), "Phase": np.angle(spec)}), output)
    click.echo(f"FFT saved to '{Path(output).name}'.")


@dsp_cmd.command("ifft")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--output", required=True, type=click.Path())
@click.option("--n", type=int, default=None, help="Length of the output signal.")
def dsp_ifft(input_file, output, n):
    """Compute the Inverse Fast Fourier Transform (IFFT)."""
    from ..core.dsp import compute_ifft
    res = sio.read_data(input_file)
    if isinstance(res, dict) and "spectrum" in res:
        spec = np.asarray(res["spectrum"])
    elif isinstance(res, pd.DataFrame) and {"Magnitude", "Phase"} <= set(res.columns):
        spec = res["Magnitude"].to_numpy() * np.exp(1j * res["Phase"].to_numpy())
    else:
        raise click.UsageError("Input must be an NPZ with 'spectrum' or a CSV with Magnitude and Phase columns.")
    try:
        x = compute_ifft(spec, n=n)
    except ValueError as e:
        raise click.UsageError(str(e))
    sio.save_data(x, output)
    click.echo(f"IFFT saved to '{Path(output).name}'.")


@dsp_cmd.command("psd-welch")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--output", required=True, type=click.Path())
@click.option("--fs", type=float, required=True)
@click.option("--window", default="hann", show_default=True)
@click.option("--nperseg", type=int, default=None)
@click.option("--noverlap", type=int, default=None)
@click.option("--nfft", type=int, default=None)
@click.option("--detrend", type=click.Choice(["none", "constant", "linear"]), default="constant", show_default=True)
@click.option("--scaling", type=click.Choice(["density", "spectrum"]), default="density", show_default=True)
def dsp_welch(input_file, output, fs, window, nperseg, noverlap, nfft, detrend, scaling):
    """Estimate Power Spectral Density using Welch's method."""
    from ..core.dsp import compute_psd_welch
    try:
        x, _ = _load_signal(input_file, fs)
        f, p = compute_psd_welch(x, fs=fs, window=window, nperseg=nperseg, noverlap=noverlap, nfft=nfft,
                                 detrend=False if detrend == "none" else detrend, scaling=scaling)
    except ValueError as e:
        raise click.UsageError(str(e))
    if Path(output).suffix.lower() == ".npz":
        sio.save_data({"frequencies": f, "psd": p}, output)
    else:
        sio.save_data(pd.DataFrame({"Frequency": f, "PSD": p}), output)
    click.echo(f"Welch PSD saved to '{Path(output).name}'.")


@dsp_cmd.command("psd-periodogram")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--output", required=True, type=click.Path())
@click.option("--fs", type=float, required=True)
@click.option("--window", default="hann", show_default=True)
@click.option("--nfft", type=int, default=None)
@click.option("--detrend", type=click.Choice(["none", "constant", "linear"]), default="constant", show_default=True)
@click.option("--scaling", type=click.Choice(["density", "spectrum"]), default="density", show_default=True)
def dsp_periodogram(input_file, output, fs, window, nfft, detrend, scaling):
    """Estimate Power Spectral Density using Periodogram."""
    from ..core.dsp import compute_psd_periodogram
    try:
        x, _ = _load_signal(input_file, fs)
        f, p = compute_psd_periodogram(x, fs=fs, window=window, nfft=nfft,
                                       detrend=False if detrend == "none" else detrend, scaling=scaling)
    except ValueError as e:
        raise click.UsageError(str(e))
    if Path(output).suffix.lower() == ".npz":
        sio.save_data({"frequencies": f, "psd": p}, output)
    else:
        sio.save_data(pd.DataFrame({"Frequency": f, "PSD": p}), output)
    click.echo(f"Periodogram PSD saved to '{Path(output).name}'.")


def _save_series(y, sr, output):
    if Path(output).suffix.lower() == ".wav":
        if sr is None:
            raise click.UsageError("Writing audio needs a sampling rate; the input file carries none.")
        sio.save_data((y, int(sr)), output)
    else:
        sio.save_data(y, output)


@dsp_cmd.command("convolution")
@click.argument("input_file_1", type=click.Path(exists=True, dir_okay=False))
@click.argument("input_file_2", type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--output", required=True, type=click.Path())
@click.option("--mode", type=click.Choice(["full", "valid", "same"]), default="same", show_default=True)
def dsp_convolution(input_file_1, input_file_2, output, mode):
    """Apply convolution to a 1D signal using a 1D kernel."""
    from ..core.dsp import apply_convolution
    x, sr = _load_signal(input_file_1, None)
    k, _ = _load_signal(input_file_2, None)
    try:
        y = apply_convolution(x, k, mode=mode)
    except ValueError as e:
        raise click.UsageError(str(e))
    _save_series(y, sr if mode == "same" else None, output)
    click.echo(f"Convolution ({mode}) saved to '{Path(output).name}'.")


@dsp_cmd.command("correlation")
@click.argument("input_file_1", type=click.Path(exists=True, dir_okay=False))
@click.argument("input_file_2", type=click.Path(exists=True, dir_okay=False), required=False)
@click.option("-o", "--output", required=True, type=click.Path())
@click.option("--mode", type=click.Choice(["full", "valid", "same"]), default="full", show_default=True)
@click.option("--method", type=click.Choice(["auto", "direct", "fft"]), default="auto", show_default=True)
def dsp_correlation(input_file_1, input_file_2, output, mode, method):
    """Compute cross-correlation (two inputs) or autocorrelation (one input)."""
    from ..core.dsp import compute_autocorrelation, compute_correlation
    x, _ = _load_signal(input_file_1, None)
    try:
        if input_file_2 is None:
            y = compute_autocorrelation(x, mode=mode, method=method)
        else:
            y = compute_correlation(x, _load_signal(input_file_2, None)[0], mode=mode, method=method)
    except ValueError as e:
        raise click.UsageError(str(e))
    _save_series(y, None, output)
    click.echo(f"{'Auto' if input_file_2 is None else 'Cross-'}correlation saved to '{Path(output).name}'.")


@dsp_cmd.command("hilbert")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--output", required=True, type=click.Path())
def dsp_hilbert(input_file, output):
    """Compute the Analytic Signal using the Hilbert Transform (output is complex)."""
    from ..core.transforms import hilbert_transform
    x, _ = _load_signal(input_file, None)
    try:
        a = hilbert_transform(x)
    except ValueError as e:
        raise click.UsageError(str(e))
    if Path(output).suffix.lower() == ".npz":
        sio.save_data({"analytic_signal": a, "envelope": np.abs(a)}, output)
    else:
        sio.save_data(pd.DataFrame({"Real": a.real, "Imag": a.imag, "Envelope": np.abs(a)}), output)
    click.echo(f"Analytic signal saved to '{Path(output).name}'.")


@dsp_cmd.command("stft")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False))
@click.option("-o", "--output", required=True, type=click.Path())
@click.option("--n-fft", type=int, default=2048, show_default=True)
@click.option("--hop-length", type=int, default=None)
@click.option("--window", default="hann", show_default=True)
def dsp_stft(input_file, output, n_fft, hop_length, window):
    """Compute the Short-Time Fourier Transform (saved as NPZ: stft, n_fft, hop_length)."""
    from ..core.dsp import compute_stft
    x, _ = _load_signal(input_file, None)
    try:
        X = compute_stft(x, n_fft=n_fft, hop_length=hop_length, window=window)
    except ValueError as e:
        raise click.UsageError(str(e))
    sio.save_data({"stft": X, "n_fft": np.array(n_fft), "hop_length": np.array(hop_length or n_fft // 4)}, output)
    click.echo(f"STFT saved to '{Path(output).name}'.")


# ---------------------------------------------------------------- filter
@click.group("filter")
def filter_cmd():
    """Apply Butterworth filters (zero-phase)."""


@filter_cmd.command("apply")
@click.argument("input_file", type=click.Path(exists=True, dir_okay=False))
@click.option("--type", "ftype", required=True, type=click.Choice(["lowpass", "highpass", "bandpass", "bandstop"]))
@click.option("--cutoff", required=True, help="Cutoff (Hz); comma-separated pair for bandpass/bandstop.")
@click.option("--fs", type=float, default=None, help="Sampling frequency (Hz) if the file carries none.")
@click.option("--order", type=int, default=5, show_default=True)
@click.option("-o", "--output", required=True, type=click.Path())
def filter_apply(input_file, ftype, cutoff, fs, order, output):
    """Apply a Butterworth filter to a signal or audio file. Uses zero-phase filtering."""
    from ..core.filters import apply_sos_filter, design_butterworth_sos
    x, sr = _load_signal(input_file, fs)
    if sr is None:
        raise click.UsageError("--fs is required: the input file carries no sampling rate.")
    parts = [float(c) for c in str(cutoff).split(",")]
    cut = parts[0] if len(parts) == 1 else (parts[0], parts[1])
    try:
        y = apply_sos_filter(design_butterworth_sos(cut, sr, order, ftype), x)
    except (ValueError, TypeError) as e:
        raise click.UsageError(str(e))
    if Path(output).suffix.lower() == ".wav":
        sio.save_data((y, int(sr)), output)
    else:
        sio.save_data(y, output)
    click.echo(f"Filtered signal saved to '{Path(output).name}'.")


cli.add_command(features_cmd)
cli.add_command(dsp_cmd)
cli.add_command(filter_cmd)

if __name__ == "__main__":
    cli()
