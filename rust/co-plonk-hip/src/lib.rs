//! MI355X-backed implementors of co-plonk's driver seam `CircomPlonkProver<P>` (co-circom/co-plonk/src/mpc.rs:16-185) for the plain,
//! Rep3 and Shamir protocols. The four data-parallel methods -- `local_mul_vec` (:57), `fft` (:138), `ifft` (:144) and
//! `msm_public_points_g1` (:164) -- land in `libcosnarks_hip.so`; the other nineteen are generated delegations to the reference's own
//! drivers (`cold.rs`, tools/gen_rust_delegates.py), so shares, states, network rounds and transcripts are untouched. The driver is
//! a type parameter of `Plonk<P, T>` (co-plonk/src/lib.rs), hence a one-identifier change at the call site:
//!
//! ```ignore
//! type HipRep3CoPlonk<P> = co_plonk::Plonk<P, co_plonk_hip::HipRep3PlonkDriver>;
//! ```
mod cold;
pub mod drivers;
pub use drivers::{HipPlainPlonkDriver, HipRep3PlonkDriver, HipShamirPlonkDriver};
