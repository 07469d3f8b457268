"""Reference-named shim: put this directory on sys.path and the reference's own
imports (`from SimpleDecoder_TransformerOnly import Decoder`) resolve to the MI355X implementation."""
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from ast_amd.SimpleDecoder_TransformerOnly import *  # noqa: F401,F403,E402
from ast_amd.SimpleDecoder_TransformerOnly import Decoder, compute_comprehensive_loss  # noqa: F401,E402
