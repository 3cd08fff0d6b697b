//! MI355X-backed implementors of co-groth16's two seams.
//!
//! * [`HipCircomReduction`] / [`HipLibSnarkReduction`]: `R1CSToQAP` (co-circom/co-groth16/src/groth16/reduction.rs:27-36).
//! * [`HipPlainGroth16Driver`] / [`HipRep3Groth16Driver`] / [`HipShamirGroth16Driver`]: `CircomGroth16Prover<P>`
//!   (co-circom/co-groth16/src/mpc.rs:22-138).
//!
//! Every hot call lands in `libcosnarks_hip.so` through `cosnarks-hip-sys`; every other trait method is delegated to the
//! reference's own driver, so the protocol logic (network rounds, share semantics) is untouched.
pub mod bases;
pub mod domain;
pub mod drivers;
pub mod error;
pub mod hip_reduction;
pub mod layout;
pub mod matrices;
pub mod split;

pub use drivers::{HipPlainGroth16Driver, HipRep3Groth16Driver, HipShamirGroth16Driver};
pub use hip_reduction::{HipCircomReduction, HipLibSnarkReduction};

use co_groth16::CoGroth16;

/// Same shape as the aliases at co-circom/co-groth16/src/groth16.rs:47-52.
pub type HipGroth16<P> = CoGroth16<P, HipPlainGroth16Driver>;
pub type HipRep3CoGroth16<P> = CoGroth16<P, HipRep3Groth16Driver>;
pub type HipShamirCoGroth16<P> = CoGroth16<P, HipShamirGroth16Driver>;

/// Bind the calling thread (a rayon worker, a party thread) to a GPU. Handles are per device; party p of an in-process
/// three-party test uses device p % count (BASELINE config 4: one GPU per party).
pub fn bind_device(device: i32) -> eyre::Result<()> {
    error::check(unsafe { cosnarks_hip_sys::csh_init(device) })
}
