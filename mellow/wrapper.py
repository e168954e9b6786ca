"""`from mellow.wrapper import MellowWrapper` (the reference's module path) -> the MI355X engine's wrapper."""
from mellow_amd.wrapper import MellowWrapper  # noqa: F401
