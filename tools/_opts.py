"""Developer tools: engine options from the command line.  `--opt KEY=VALUE` (repeatable, anywhere on the line) is removed from
sys.argv and returned as the `options=` dict of mellow_amd.engine.Engine (mellow_engine_set_option): the library reads no
environment variable, so the A/B forms the tools compare are named here, e.g.
    python tools/decode_probe.py 64 64 --opt decode_x3=0
    python tools/pmc_prefill.py --opt prefill_split=1"""
import sys


def engine_options():
    out, keep, it = {}, [], iter(sys.argv)
    for a in it:
        if a == "--opt":
            k, v = next(it).split("=", 1)
            out[k] = int(v, 0)
        elif a.startswith("--opt="):
            k, v = a[6:].split("=", 1)
            out[k] = int(v, 0)
        else:
            keep.append(a)
    sys.argv[:] = keep
    return out
