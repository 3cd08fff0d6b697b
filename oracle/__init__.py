"""CPU restatement (oracle) of the co-snarks proof-generation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or
executed by the product path (``co-snarks_amd/``); only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it, and
there only as the checker.

Two layers:

* pure-Python big-integer restatement (``fields``, ``curves``, ``msm``, ``ntt``,
  ``mpc``, ``zkey``, ``pairing``, ``groth16``) -- small cases, golden-vector
  generation, pinned against the reference's own ``test_vectors/Groth16``
  fixtures (committed snarkjs proofs verify; proofs produced by the restated
  prover verify) and against the BN254 Fr multiplication known-answer test of
  ``tests/tests/mpc/rep3.rs:286-345``.
* plain-C restatement (``oracle/c/``) -- same algorithms at scale, pinned
  against the Python layer and the committed fixtures; also the timed
  ``cpu_baseline`` ("port") of ``bench.py``.

The MSM/FFT arithmetic of the reference lives in the un-vendored crates
``taceo-ark-algebra 0.1.0`` / ``ark-ec|ff|poly 0.6.0`` (``Cargo.toml:84``,
``Cargo.lock:4771-4781``); this oracle restates the published algorithms and
anchors parity on the reference's call sites and fixtures.
"""
