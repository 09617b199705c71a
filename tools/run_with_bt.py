"""python tools/run_with_bt.py <pytest args>: pytest with a native backtrace printed on SIGSEGV (debug aid)."""
import ctypes, os, sys
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "libsegv_bt.so"))
import pytest
import torch
torch.cuda.init()
torch.zeros(1, device='cuda')
import faulthandler
faulthandler.disable()
lib.dm_install_segv_bt()
sys.exit(pytest.main(sys.argv[1:] + ["-p", "no:faulthandler"]))
