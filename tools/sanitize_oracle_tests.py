import sys, ctypes as C, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import _oracle
orig = _oracle.load
def load(native=False):
    lib = C.CDLL(os.path.join(os.environ["EARHIP_SANITIZE_DIR"], "liboracle.so"))
    lib.oracle_last_error.restype = C.c_char_p
    for fn in ("oracle_conv_ctx_create","oracle_conv_filter_create","oracle_delay_create","oracle_vbs_create","oracle_render_create"):
        getattr(lib, fn).restype = C.c_void_p
    lib.oracle_conv_filter_num_blocks.restype = C.c_size_t
    return lib
_oracle.load = load
import pytest
rc = pytest.main(["-q","tests/test_oracle_fft.py","tests/test_oracle_gain_interp.py","tests/test_oracle_block_convolver.py","tests/test_oracle_delay_vbs.py","tests/test_oracle_decorrelate.py","-p","no:cacheprovider"])
print("RC", rc, flush=True)
os._exit(int(rc))
