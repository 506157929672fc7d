"""GEMM solution selection for the dense contractions (PyTorch TunableOp over rocBLAS + hipBLASLt).

The 1x1 convolutions, the DCN contraction and the R-FCN score-map GEMM are skinny fp32 GEMMs
(2394 or 37500 rows, 64..4608 deep) for which the libraries' default heuristics pick tiles that run
at a fraction of what their best solution does (measured at 1000x600: conv3 of a stage-3 unit,
(2394 x 256) x (256 x 1024) accumulated into the shortcut, 35 us with the default, 16 us with the
tuned solution; whole backbone 5.18 ms -> 4.30 ms).  `lsfa_amd/tuned/gemm_gfx950.csv` holds the
solutions found on an MI355X for the shapes of the 1000x600 workload (tools/tune_gemms.py writes
it); `enable()` loads it.  Shapes that are not in the file (another resolution) are tuned on first
use when `tune_missing` is set — a few seconds per shape, during warm-up, never inside a captured
graph — and kept in memory only: the shipped file is never written at run time.  If the library
versions recorded in the file do not match the running ones, PyTorch rejects the file and the same
on-first-use tuning applies.
"""
import os
import tempfile

import torch

SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tuned', 'gemm_gfx950.csv')


def enable(tune_missing=True, results_file=None):
    """Turn TunableOp on for this process.  Returns True when the shipped results were accepted."""
    import torch.cuda.tunable as T
    T.enable(True)
    # new results go to a scratch file of this process, not into the package
    T.set_filename(results_file or os.path.join(tempfile.gettempdir(), 'lsfa_tunableop_%d.csv' % os.getpid()))
    ok = False
    if results_file is None and os.path.exists(SHIPPED):
        try:
            ok = bool(T.read_file(SHIPPED))
        except RuntimeError:
            ok = False
    T.tuning_enable(bool(tune_missing))
    return ok


def disable():
    import torch.cuda.tunable as T
    T.tuning_enable(False)
    T.enable(False)


_PRIVATE_DIR = None


def pin_algorithms(isolate_miopen=False):
    """Deterministic library algorithm choice (`python -m lsfa_amd.test --pinned-algorithms`, the multi-rank equality
    test): no TunableOp, no MIOpen find step.  What is left of MIOpen's state dependence, measured
    (tools/diag_multirank.py, profiles/r3/multirank_diag_before.txt): with `cudnn.benchmark` off MIOpen still picks a
    convolution's solver from per-user state under $HOME; two processes that START TOGETHER on an empty cache end up with
    other solvers for FlowNet than a process that starts alone (frames from the first FlowNet key frame on move by up to
    4e-5 px), and that choice persists in $HOME.  Processes that find the state already written agree bit for bit.
    isolate_miopen=True points MIOPEN_USER_DB_PATH / MIOPEN_CUSTOM_CACHE_DIR at a fresh directory (it did NOT make the
    first process of a box agree with later ones, so it is off by default); the way out is the own convolution for
    FlowNet.  Must run before the first convolution of the process."""
    global _PRIVATE_DIR
    disable()
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False
    if isolate_miopen and _PRIVATE_DIR is None:
        import atexit
        import shutil
        _PRIVATE_DIR = tempfile.mkdtemp(prefix='lsfa_miopen_%d_' % os.getpid())
        os.environ['MIOPEN_USER_DB_PATH'] = os.path.join(_PRIVATE_DIR, 'db')
        os.environ['MIOPEN_CUSTOM_CACHE_DIR'] = os.path.join(_PRIVATE_DIR, 'cache')
        for d in ('db', 'cache'):
            os.makedirs(os.path.join(_PRIVATE_DIR, d), exist_ok=True)
        atexit.register(shutil.rmtree, _PRIVATE_DIR, True)
    return _PRIVATE_DIR
