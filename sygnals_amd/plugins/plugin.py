"""sygnals plugin exposing the device backend through the reference's plugin API.

Subclasses sygnals.plugins.api.SygnalsPluginBase (sygnals/plugins/api.py:37-138) when the
reference package is importable and registers the backend's callables through
PluginRegistry.add_filter / add_transform / add_feature / add_cli_command (api.py:173-246)
under the reference's own function names.  When `sygnals` is not installed (e.g. on a bare GPU
box) a structurally identical local base class is used so the plugin can still be
instantiated and exercised against any object implementing the registry methods.

Discovery: entry-point group ``sygnals.plugins`` (loader.py:32, 309) -- see pyproject.toml --
or a copy of this directory (with plugin.toml) under the configured plugin_dir
(loader.py:355-372).
"""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from typing import Any, Dict

logger = logging.getLogger(__name__)

try:  # the real base class when the reference package is present
    from sygnals.plugins.api import SygnalsPluginBase as _Base  # type: ignore
except Exception:  # pragma: no cover - exercised on boxes without the reference
    class _Base(ABC):
        """Hook-compatible stand-in for SygnalsPluginBase (same hook names and call order)."""

        @property
        @abstractmethod
        def name(self) -> str: ...

        @property
        @abstractmethod
        def version(self) -> str: ...

        def register_filters(self, registry): pass
        def register_transforms(self, registry): pass
        def register_feature_extractors(self, registry): pass
        def register_visualizations(self, registry): pass
        def register_audio_effects(self, registry): pass
        def register_augmenters(self, registry): pass
        def register_data_readers(self, registry): pass
        def register_data_writers(self, registry): pass
        def register_cli_commands(self, registry): pass
        def setup(self, config: Dict[str, Any]): pass
        def teardown(self): pass


class SygnalsAmdPlugin(_Base):
    @property
    def name(self) -> str:
        return "sygnals-amd"

    @property
    def version(self) -> str:
        from .. import __version__
        return __version__

    # ---- lifecycle ------------------------------------------------------------------
    def setup(self, config: Dict[str, Any]):
        """Load the HIP library eagerly so a missing build fails at plugin load, not mid-run."""
        from .._lib import lib
        lib()
        self._config = dict(config or {})
        logger.debug("sygnals-amd: libsygnals_hip.so loaded")

    def teardown(self):
        from .. import ops
        ops._dev_cache.clear()
        ops._mfcc_calls.clear()

    # ---- registration hooks (call order fixed by loader.py:266-274) -----------------------
    def register_filters(self, registry):
        from ..core import filters as F
        for fn in (F.apply_sos_filter, F.low_pass_filter, F.high_pass_filter, F.band_pass_filter,
                   F.band_stop_filter):
            registry.add_filter(fn.__name__, fn)
        registry.add_filter("apply_sos_filter_batch", F.apply_sos_filter_batch)

    def register_transforms(self, registry):
        from ..core import dsp as D
        from ..core import transforms as TR
        for fn in (D.compute_fft, D.compute_ifft, D.compute_stft, D.compute_cqt, D.compute_psd_welch, D.apply_window,
                   D.apply_convolution, D.compute_correlation, D.compute_autocorrelation, D.compute_psd_periodogram,
                   D.amplitude_envelope, TR.hilbert_transform):
            registry.add_transform(fn.__name__, fn)

    def register_feature_extractors(self, registry):
        from ..core.audio import features as af
        from ..core.features import cepstral, frequency_domain as fd, manager, time_domain as td
        for name, fn in fd.FREQUENCY_DOMAIN_FEATURES.items():
            registry.add_feature(name, fn)
        for name, fn in td.TIME_DOMAIN_FEATURES.items():
            registry.add_feature(name, fn)
        registry.add_feature("zero_crossing_rate", af.zero_crossing_rate)
        registry.add_feature("rms_energy", af.rms_energy)
        registry.add_feature("spectral_contrast", fd.spectral_contrast)
        registry.add_feature("mfcc", cepstral.mfcc)
        registry.add_feature("extract_features", manager.extract_features)
        registry.add_feature("extract_features_batch", manager.extract_features_batch)

    def register_cli_commands(self, registry):
        from ..cli.main import dsp_cmd, features_cmd, filter_cmd
        for cmd in (features_cmd, dsp_cmd, filter_cmd):
            registry.add_cli_command(cmd)
