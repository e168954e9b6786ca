"""Drop-in import name of the reference package (reference mellow/__init__.py:1, README.md:47):

    from mellow import MellowWrapper

resolves to the MI355X engine's wrapper (`mellow_amd.MellowWrapper`), so the reference's `example.py` flow runs
unchanged.  Everything lives in `mellow_amd`; this package only forwards the public names."""
__all__ = ["MellowWrapper"]


def __getattr__(name):
    if name == "MellowWrapper":
        from mellow_amd.wrapper import MellowWrapper
        return MellowWrapper
    raise AttributeError(name)
