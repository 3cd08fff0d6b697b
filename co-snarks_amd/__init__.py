"""co-snarks_amd -- MI355X (gfx950) proof-generation hot path for TaceoLabs/co-snarks.

The product is two native libraries built in-tree by ``build.py``:

* ``lib/libcosnarks_hip.so``      -- the C ABI declared in ``include/cosnarks_hip.h`` (hand-written HIP:
  Pippenger MSM, LDS-tiled radix-2 NTT, Rep3/Shamir share-vector arithmetic);
* ``lib/libcosnarks_groth16.so``  -- C++ host mirror of the reference's ``CircomGroth16Prover`` /
  ``R1CSToQAP`` / ``CoGroth16::prove`` interface, calling only that C ABI.

This Python package is a thin ctypes harness over them (tests, bench, smoke).  There is NO CPU fallback:
every compute call raises ``CoSnarksHipError`` if the library is missing or no HIP device is present.
The directory name contains a hyphen; import it as ``import cosnarks_amd`` (alias module at the repo root)
or ``importlib.import_module("co-snarks_amd")``.
"""
from . import bindings  # noqa: F401
from .bindings import (  # noqa: F401
    BLS12_381,
    BN254,
    GRUMPKIN,
    G1,
    G2,
    Bases,
    CoSnarksHipError,
    Comm,
    DeviceBuffer,
    Domain,
    device_count,
    fr_bytes,
    groth16_h,
    have_device,
    lib,
    lib_path,
    lincomb,
    msm_fold_partials,
    msm_partial_bytes,
    msm_split,
    point_bytes,
    rep3_local_mul_vec,
    rep3_to_shamir_vec,
    tune_get,
    msm_plan,
    tune_set,
    tuned,
    vec_add,
    vec_mul,
    vec_mul_table,
    vec_sub,
)
