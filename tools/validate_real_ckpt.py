#!/usr/bin/env python3
"""Launcher: `python tools/validate_real_ckpt.py ...` == `python tests/validate_real_ckpt.py ...` (the validation drives the CPU
oracle, which only code under tests/ may do; see that file's docstring for the arguments).  Needs MELLOW_CKPT_DIR + the tokenizer
files, or --synthetic."""
import os
import runpy
import sys

sys.exit(runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "validate_real_ckpt.py"),
                        run_name="__main__"))
