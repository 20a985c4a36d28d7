from .plugin import SygnalsAmdPlugin  # noqa: F401
