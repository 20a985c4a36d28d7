"""Plugin registration through the reference's registry API and the CLI surface (CPU only)."""
import os
import sys

import numpy as np
import pytest
from click.testing import CliRunner

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeRegistry:
    """Implements the PluginRegistry methods the plugin calls (sygnals/plugins/api.py:173-246)."""
    def __init__(self):
        self.filters, self.transforms, self.features, self.cli = {}, {}, {}, []
    def add_filter(self, n, f): self.filters[n] = f
    def add_transform(self, n, f): self.transforms[n] = f
    def add_feature(self, n, f): self.features[n] = f
    def add_cli_command(self, c): self.cli.append(c)


def _run_hooks(plugin, reg):
    # hook order of sygnals/plugins/loader.py:266-274
    plugin.setup({})
    for hook in ("register_filters", "register_transforms", "register_feature_extractors", "register_visualizations",
                 "register_audio_effects", "register_augmenters", "register_data_readers", "register_data_writers",
                 "register_cli_commands"):
        getattr(plugin, hook)(reg)


def _check_registered(filters, transforms, features, cli_names):
    assert {"apply_sos_filter", "low_pass_filter", "high_pass_filter", "band_pass_filter", "band_stop_filter"} <= set(filters)
    assert {"compute_fft", "compute_ifft", "compute_stft", "compute_psd_welch", "apply_window", "apply_convolution",
            "compute_correlation", "compute_autocorrelation", "compute_psd_periodogram", "amplitude_envelope",
            "hilbert_transform"} <= set(transforms)
    assert {"spectral_centroid", "spectral_bandwidth", "spectral_flatness", "spectral_rolloff", "dominant_frequency",
            "spectral_contrast", "mfcc", "extract_features"} <= set(features)
    assert {"features", "dsp", "filter"} <= set(cli_names)


def test_plugin_registers_into_fake_registry():
    from sygnals_amd.plugins import SygnalsAmdPlugin
    p, reg = SygnalsAmdPlugin(), FakeRegistry()
    assert p.name == "sygnals-amd" and p.version
    _run_hooks(p, reg)
    _check_registered(reg.filters, reg.transforms, reg.features, [c.name for c in reg.cli])
    p.teardown()


def test_plugin_against_the_reference_registry_when_present():
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "sygnals", "plugins")):
        pytest.skip("reference checkout not present on this box")
    sys.path.insert(0, ref)
    try:
        from sygnals.plugins.api import PluginRegistry, SygnalsPluginBase
        import importlib
        import sygnals_amd.plugins.plugin as mod
        mod = importlib.reload(mod)
        p, reg = mod.SygnalsAmdPlugin(), PluginRegistry()
        assert isinstance(p, SygnalsPluginBase)
        _run_hooks(p, reg)
        _check_registered(reg.list_filters(), reg.list_transforms(), reg.list_features(),
                          [c.name for c in reg.get_cli_commands()])
        assert reg.get_feature("mfcc") is not None and reg.get_filter("band_pass_filter") is not None
    finally:
        sys.path.remove(ref)


def test_manifest_has_the_required_fields():
    import tomli
    m = tomli.load(open(os.path.join(ROOT, "sygnals_amd", "plugins", "plugin.toml"), "rb"))
    assert {"name", "version", "sygnals_api", "entry_point"} <= set(m)       # loader.py:72
    from packaging.specifiers import SpecifierSet
    from packaging.version import Version
    assert Version("1.0.0") in SpecifierSet(m["sygnals_api"])                # loader.py:85-103, version.py:8
    mod, cls = m["entry_point"].split(":")
    import importlib
    assert hasattr(importlib.import_module(mod), cls)


def test_cli_surface():
    from sygnals_amd.cli.main import cli
    r = CliRunner()
    out = r.invoke(cli, ["features", "extract", "--help"]).output
    for opt in ("-o, --output", "-f, --feature", "--frame-length", "--hop-length", "2048", "512"):
        assert opt in out
    out = r.invoke(cli, ["dsp", "--help"]).output
    for c in ("fft", "ifft", "psd-welch", "stft", "psd-periodogram", "convolution", "correlation", "hilbert"):
        assert c in out
    out = r.invoke(cli, ["dsp", "correlation", "--help"]).output          # README.md:768-792
    for opt in ("--mode", "--method", "[INPUT_FILE_2]", "-o, --output"):
        assert opt in out
    out = r.invoke(cli, ["dsp", "psd-periodogram", "--help"]).output      # README.md:795-815
    for opt in ("--fs", "--window", "--nfft", "--detrend", "--scaling"):
        assert opt in out
    out = r.invoke(cli, ["filter", "apply", "--help"]).output
    for opt in ("--type", "--cutoff", "--fs", "--order", "-o, --output"):
        assert opt in out
    res = r.invoke(cli, ["features", "extract", "/nonexistent.wav", "-o", "x.npz", "-f", "mfcc"])
    assert res.exit_code == 2                                                # click usage error


def test_io_roundtrip(tmp_path):
    from scipy.io import wavfile
    from sygnals_amd import io as sio
    import pandas as pd
    x = (np.sin(np.arange(1600) * 0.1) * 20000).astype(np.int16)
    wavfile.write(str(tmp_path / "a.wav"), 16000, x)
    y, sr = sio.read_data(tmp_path / "a.wav")
    assert sr == 16000 and y.dtype == np.float64 and np.allclose(y, x / 32768.0)
    st = np.stack([x, x // 2], axis=1)
    wavfile.write(str(tmp_path / "s.wav"), 16000, st)
    y2, _ = sio.read_data(tmp_path / "s.wav")
    assert y2.shape == (2, 1600)
    # feature table: CSV is written WITHOUT the time index (reference data_handler.py:248), NPZ keeps 'time'
    df = pd.DataFrame({"mfcc_0": [1.0, 2.0]}, index=pd.to_timedelta([0.0, 0.1], unit="s")); df.index.name = "time"
    sio.save_data(df, tmp_path / "f.csv")
    assert list(pd.read_csv(tmp_path / "f.csv").columns) == ["mfcc_0"]
    sio.save_data({"time": np.array([0.0, 0.1]), "mfcc_0": np.array([1.0, 2.0])}, tmp_path / "f.npz")
    assert set(sio.read_data(tmp_path / "f.npz")) == {"time", "mfcc_0"}
    sio.save_data(np.arange(4.0), tmp_path / "v.csv")
    assert list(sio.read_data(tmp_path / "v.csv").columns) == ["value"]
    sio.save_data(np.arange(4.0), tmp_path / "v.npz")
    sig, _ = sio.signal_from(sio.read_data(tmp_path / "v.npz"))
    assert np.array_equal(sig, np.arange(4.0))
    with pytest.raises(ValueError):
        sio.save_data({"a": np.zeros(2)}, tmp_path / "d.csv")


def test_read_clips_batches_mixed_inputs(tmp_path):
    """Batched ingest: WAV (mono + stereo mix-down), NPZ and CSV clips into one padded [B, L] batch."""
    from sygnals_amd import io as sio
    rng = np.random.default_rng(0)
    a = rng.uniform(-0.5, 0.5, 1000)
    st = rng.uniform(-0.5, 0.5, (2, 700))
    sio.save_data((a, 16000), tmp_path / "a.wav", sr=16000)
    from scipy.io import wavfile
    wavfile.write(str(tmp_path / "st.wav"), 16000, np.round(st.T * 32767).astype(np.int16))
    np.savez(tmp_path / "n.npz", data=a[:300], sr=8000)
    sio.save_data(a[:50], tmp_path / "c.csv")
    batch, srs = sio.read_clips([tmp_path / "a.wav", tmp_path / "st.wav", tmp_path / "n.npz", tmp_path / "c.csv"])
    assert batch.dtype == np.float32 and batch.shape == (4, 1000) and srs == [16000, 16000, 8000, None]
    assert np.abs(batch[0] - np.round(a * 32767) / 32768).max() < 1e-6
    assert np.abs(batch[1, :700] - (np.round(st * 32767) / 32768).mean(axis=0)).max() < 1e-6 and not batch[1, 700:].any()
    assert np.allclose(batch[2, :300], a[:300], atol=1e-7) and np.allclose(batch[3, :50], a[:50], atol=1e-7)
    cut, _ = sio.read_clips([tmp_path / "a.wav"], length=10)
    assert cut.shape == (1, 10)
    with pytest.raises(ValueError):
        sio.read_clips([])


def test_read_clips_pcm_keeps_integer_samples(tmp_path):
    """f-2 ingest without host conversion: one integer [B, L, C] array, zero (uint8: 128) padded, cut to length."""
    from scipy.io import wavfile
    from sygnals_amd import io as sio
    rng = np.random.default_rng(1)
    a = rng.integers(-30000, 30000, (1000, 2), dtype=np.int16)
    b = rng.integers(-30000, 30000, (700, 2), dtype=np.int16)
    wavfile.write(str(tmp_path / "a.wav"), 8000, a)
    wavfile.write(str(tmp_path / "b.wav"), 8000, b)
    batch, srs = sio.read_clips_pcm([tmp_path / "a.wav", tmp_path / "b.wav"], length=800)
    assert batch.dtype == np.int16 and batch.shape == (2, 800, 2) and srs == [8000, 8000]
    assert np.array_equal(batch[0], a[:800]) and np.array_equal(batch[1, :700], b) and not batch[1, 700:].any()
    u = rng.integers(0, 255, 300, dtype=np.uint8)
    wavfile.write(str(tmp_path / "u.wav"), 8000, u)
    batch, _ = sio.read_clips_pcm([tmp_path / "u.wav"], length=400)
    assert batch.dtype == np.uint8 and batch.shape == (1, 400, 1) and (batch[0, 300:] == 128).all()
    wavfile.write(str(tmp_path / "m.wav"), 8000, a[:, 0].copy())
    with pytest.raises(ValueError, match="channel count differs"):
        sio.read_clips_pcm([tmp_path / "a.wav", tmp_path / "m.wav"])
    with pytest.raises(ValueError, match="no input files"):
        sio.read_clips_pcm([])


def test_reference_plugin_loader_discovers_and_loads_the_backend(tmp_path):
    """Drives the reference's own PluginLoader.discover_and_load() (sygnals/plugins/loader.py:291-384): a local plugin
    directory holding this package's plugin.toml -> manifest parse (:63-83), PEP 440 compatibility check against the
    core version (:85-103), entry-point import (:105-158), setup() and the nine registration hooks in loader order
    (:266-274).  Build container only (the reference does not exist on the GPU box).  The reference imports the
    `toml` package, which this image lacks: a tomli-backed module with the three names the loader uses stands in
    for it for the duration of the test."""
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "sygnals", "plugins")):
        pytest.skip("reference checkout not present on this box")
    import importlib
    import shutil
    import types
    import tomli
    stand_in = None
    try:
        import toml  # noqa: F401
    except ModuleNotFoundError:
        stand_in = types.ModuleType("toml")
        stand_in.load = lambda f: tomli.loads(f.read()) if not isinstance(f, (str, os.PathLike)) else tomli.loads(open(f).read())
        stand_in.loads = tomli.loads
        stand_in.TomlDecodeError = tomli.TOMLDecodeError
        def _dump(data, f):
            for name, d in data.items():
                f.write(f"[{name}]\n" + "".join(f"{k} = {str(v).lower() if isinstance(v, bool) else repr(v)}\n" for k, v in d.items()))
        stand_in.dump = _dump
        sys.modules["toml"] = stand_in
    sys.path.insert(0, ref)
    try:
        from sygnals.config.models import SygnalsConfig
        from sygnals.plugins.api import PluginRegistry
        from sygnals.plugins import loader as L
        import sygnals_amd.plugins.plugin as mod
        importlib.reload(mod)                                   # bind the plugin class to the reference's base class
        plug_dir = tmp_path / "plugins"
        (plug_dir / "sygnals_amd_backend").mkdir(parents=True)
        shutil.copy(os.path.join(ROOT, "sygnals_amd", "plugins", "plugin.toml"), plug_dir / "sygnals_amd_backend" / "plugin.toml")
        (plug_dir / "broken").mkdir()
        (plug_dir / "broken" / "plugin.toml").write_text('name = "broken"\nversion = "0.0.1"\n')   # missing fields: skipped
        cfg = SygnalsConfig()
        cfg.paths.plugin_dir = plug_dir
        reg = PluginRegistry()
        order = []
        for hook in ("setup", "register_filters", "register_transforms", "register_feature_extractors",
                     "register_visualizations", "register_audio_effects", "register_augmenters", "register_data_readers",
                     "register_data_writers", "register_cli_commands"):
            orig = getattr(mod.SygnalsAmdPlugin, hook)
            def wrapped(self, *a, _o=orig, _h=hook, **k):
                order.append(_h)
                return _o(self, *a, **k)
            setattr(mod.SygnalsAmdPlugin, hook, wrapped)
        try:
            ld = L.PluginLoader(cfg, reg)
            ld.discover_and_load()
        finally:
            importlib.reload(mod)
        assert list(ld.loaded_plugins) == ["sygnals-amd"] and ld.plugin_sources["sygnals-amd"] == "local"
        assert "sygnals-amd" in reg.loaded_plugin_names and "broken" not in ld.plugin_manifests
        assert order == ["setup", "register_filters", "register_transforms", "register_feature_extractors",
                         "register_visualizations", "register_audio_effects", "register_augmenters",
                         "register_data_readers", "register_data_writers", "register_cli_commands"]
        _check_registered(reg.list_filters(), reg.list_transforms(), reg.list_features(),
                          [c.name for c in reg.get_cli_commands()])
        info = {i["name"]: i for i in ld.get_plugin_info()}
        assert info["sygnals-amd"]["status"] == "loaded" and info["sygnals-amd"]["api_required"] == ">=1.0.0,<2.0.0"
        # an incompatible API requirement is refused by the loader's PEP 440 check
        (plug_dir / "sygnals_amd_backend" / "plugin.toml").write_text(
            open(os.path.join(ROOT, "sygnals_amd", "plugins", "plugin.toml")).read().replace(">=1.0.0,<2.0.0", ">=2.0.0"))
        ld2 = L.PluginLoader(cfg, PluginRegistry())
        ld2.discover_and_load()
        assert not ld2.loaded_plugins and {i["name"]: i["status"] for i in ld2.get_plugin_info()}["sygnals-amd"] == "error/incompatible"
    finally:
        sys.path.remove(ref)
        if stand_in is not None:
            sys.modules.pop("toml", None)
