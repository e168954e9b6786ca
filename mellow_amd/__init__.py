"""mellow_amd — MI355X-native inference engine for the Mellow audio-language model.

`from mellow_amd import MellowWrapper` mirrors `from mellow import MellowWrapper` of the reference."""
__all__ = ["MellowWrapper"]


def __getattr__(name):
    if name == "MellowWrapper":
        from .wrapper import MellowWrapper
        return MellowWrapper
    raise AttributeError(name)
