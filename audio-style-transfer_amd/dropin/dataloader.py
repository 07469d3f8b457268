"""Reference-named shim: put this directory on sys.path and the reference's own
imports (`from dataloader import get_dataloader`) resolve to the MI355X implementation."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from ast_amd.dataloader import *  # noqa: F401,F403,E402
from ast_amd.dataloader import normalize, concat_stft_cqt, DualInstrumentDataset, custom_collate_fn, get_dataloader  # noqa: F401,E402
