#!/usr/bin/env python3
"""Command-line front of miniprot_amd/synth.py (the deterministic genome/protein generator of SURVEY.md 8(d))."""
import os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from miniprot_amd.synth import *  # noqa: F401,F403  (tools and tests import the generator under this name)

if __name__ == "__main__":
    runpy.run_module("miniprot_amd.synth", run_name="__main__")
