"""MI355X-native layout-token training step (host side).  Compute lives in ../libvlg_hip.so."""
from .spec import LayoutConfig  # noqa: F401
