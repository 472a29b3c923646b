"""MI355X-native negacyclic-NTT / RNS polynomial-arithmetic engine.

Hot path of Stardust-lf/fhe-reliability-gpu (NTT, INTT, modmul, base conversion,
four-step) as hand-written gfx950 HIP kernels behind a C ABI
(``include/fhe_mi355x.h`` -> ``libfhe_mi355x.so``), with this thin ctypes host.
Importing the package loads the shared object and raises ImportError when it has
not been built -- there is no CPU fallback.
"""
from . import _lib  # noqa: F401  (fails loudly when the HIP extension is missing)
from ._lib import FheError  # noqa: F401
from .engine import (  # noqa: F401
    Abft,
    BaseConv,
    KeySwitch,
    automorphism,
    DeviceArray,
    Engine,
    NttTables,
    bConv,
    base_conv_fixed,
    create_moduli,
    crt_garner,
    default_engine,
    diag_block_hadamard_matvec,
    four_step_ntt,
    intt,
    intt_nthroot,
    min_primitive_root,
    negacyclic_intt,
    negacyclic_ntt,
    ntt,
    ntt_nthroot,
    poly_mul_negacyclic_ntt,
    root_powers,
)

__version__ = "0.1.0"
