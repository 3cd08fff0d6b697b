//! MI355X-backed implementors of co-noir's driver seam `NoirUltraHonkProver<P>` (co-noir/co-noir-common/src/mpc/mod.rs:17-408) for the
//! plain, Rep3 and Shamir protocols, and [`hip_fast_msm`], the GPU counterpart of `HonkCurve::fast_msm`
//! (co-noir-common/src/honk_curve.rs:35, 81-83 BN254 G1, 175-177 Grumpkin). Hot methods: `local_mul_vec` (:236), `msm_public_points`
//! (:358), `fft` (:373), `ifft` (:379); the other fifty are generated delegations to the reference's drivers (`cold.rs`).
//!
//! `HonkCurve::fast_msm` is an associated function of the curve types themselves, so call sites inside co-noir that name
//! `P::fast_msm` directly (commitments of the plain prover) stay on the CPU unless they are pointed at [`hip_fast_msm`]; every MSM
//! that goes through the driver trait (`T::msm_public_points`, all of the MPC provers' commitments) is offloaded by the implementors
//! here with no upstream edit.
mod cold;
pub mod drivers;
pub use drivers::{hip_fast_msm, HipPlainUltraHonkDriver, HipRep3UltraHonkDriver, HipShamirUltraHonkDriver};
