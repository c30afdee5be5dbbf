"""lseg_hip -- host side of the MI355X-native LSeg forward engine."""
from .config import LSegConfig, TextConfig, get_config, available_backbones  # noqa: F401
