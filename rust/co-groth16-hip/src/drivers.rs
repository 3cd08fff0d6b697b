//! `CircomGroth16Prover<P>` implementors (co-circom/co-groth16/src/mpc.rs:22-138) whose three hot methods run on the GPU.
//! Everything else -- `rand`, the constraint-row kernels, the point helpers and every network round -- is the reference's own
//! driver, reached by delegation: the associated types are the reference's, so shares, states and wire traffic are unchanged.
use crate::bases;
use crate::error::hip_ok;
use crate::layout::{curve_id, limbs_mut, limbs_of};
use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_ec::CurveGroup;
use co_groth16::mpc::{CircomGroth16Prover, PlainGroth16Driver, Rep3Groth16Driver, ShamirGroth16Driver};
use core::mem::MaybeUninit;
use cosnarks_hip_sys as sys;
use mpc_core::protocols::rep3::{Rep3PrimeFieldShare, Rep3State};
use mpc_core::protocols::shamir::{ShamirPrimeFieldShare, ShamirState};
use mpc_core::MpcState;
use mpc_net::Network;

/// What a witness map (hip_reduction.rs) asks of the next `T::local_mul_vec` on the CALLING thread. The Rep3 generators are private
/// (`Rep3Rand { rng1, rng2 }`, rngs.rs:83-86) and `R1CSToQAP::witness_map_from_matrices` is generic over `T`, so the only way to the
/// party's randomness is through `T`'s own methods; the request travels in a thread-local and any driver but the Hip Rep3 one ignores it.
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
pub enum MaskRequest {
    /// an ordinary product
    None,
    /// return the mask vector of `n` entries a product of two n-vectors would add, and nothing else: `masking_field_elements_vec(n)`
    /// (rngs.rs:137-156). The operands are EMPTY vectors -- no zero vectors of n shares are built to be thrown away (round 4 did:
    /// 4 x 64 MB of element-wise `vec![default; n]` per mask at 2^20).
    MaskOnly(usize),
    /// draw one pair of fresh correlated seeds, `Rep3Rand::random_seeds()` (rngs.rs:233, public), leave them in [`take_seeds`] and
    /// return an empty vector: the opt-in all-GPU-parties mode, the device generates both mask vectors from the seeds
    Seeds,
}
thread_local! {
    static MASK_REQUEST: core::cell::Cell<MaskRequest> = const { core::cell::Cell::new(MaskRequest::None) };
    static SEEDS: core::cell::Cell<Option<([u8; 32], [u8; 32])>> = const { core::cell::Cell::new(None) };
}

/// Run `f` (one `T::local_mul_vec` call on the CALLING thread) with `req` posted.
pub fn with_mask_request<R>(req: MaskRequest, f: impl FnOnce() -> R) -> R {
    struct Reset;
    impl Drop for Reset {
        fn drop(&mut self) {
            MASK_REQUEST.with(|m| m.set(MaskRequest::None));
        }
    }
    MASK_REQUEST.with(|m| m.set(req));
    let _reset = Reset; // also on unwind
    f()
}

/// The seeds the Hip Rep3 driver left for a `MaskRequest::Seeds` call (None: the driver in use did not honour the request).
pub fn take_seeds() -> Option<([u8; 32], [u8; 32])> {
    SEEDS.with(|s| s.take())
}

/// `msm_unchecked(points, scalars)` (external taceo-ark-algebra; call sites mpc/plain.rs:66-74, mpc/rep3.rs:124-132,
/// mpc/shamir.rs:111-119): "unchecked" = the shorter of the two lengths (honk_curve.rs:33-34).
fn hip_msm<P: Pairing, C>(points: &[Affine<C>], scalars: &[P::ScalarField]) -> Projective<C>
where
    C: SWCurveConfig<ScalarField = P::ScalarField>,
{
    let n = points.len().min(scalars.len());
    let (b, off) = bases::get_or_upload::<P, C>(points);
    let mut out = MaybeUninit::<Projective<C>>::uninit(); // Jacobian {x, y, z}: the library writes (X, Y, Z), Z in {0, 1}
    hip_ok(unsafe { sys::csh_msm(b.handle, off, n, limbs_of(scalars), 1, out.as_mut_ptr().cast()) });
    unsafe { out.assume_init() }
}

macro_rules! delegate_cold_methods {
    ($inner:ty) => {
        fn rand<N: Network>(net: &N, state: &mut Self::State) -> eyre::Result<Self::ArithmeticShare> {
            <$inner as CircomGroth16Prover<P>>::rand(net, state)
        }
        fn evaluate_constraint(
            id: <Self::State as MpcState>::PartyID,
            lhs: &[(P::ScalarField, usize)],
            public_inputs: &[P::ScalarField],
            private_witness: &[Self::ArithmeticShare],
        ) -> Self::ArithmeticShare {
            <$inner as CircomGroth16Prover<P>>::evaluate_constraint(id, lhs, public_inputs, private_witness)
        }
        fn evaluate_constraint_half_share(
            id: <Self::State as MpcState>::PartyID,
            lhs: &[(P::ScalarField, usize)],
            public_inputs: &[P::ScalarField],
            private_witness: &[Self::ArithmeticShare],
        ) -> Self::ArithmeticHalfShare {
            <$inner as CircomGroth16Prover<P>>::evaluate_constraint_half_share(id, lhs, public_inputs, private_witness)
        }
        fn promote_to_trivial_shares(
            id: <Self::State as MpcState>::PartyID,
            public_values: &[P::ScalarField],
        ) -> Vec<Self::ArithmeticShare> {
            <$inner as CircomGroth16Prover<P>>::promote_to_trivial_shares(id, public_values)
        }
        fn to_half_share(a: Self::ArithmeticShare) -> Self::ArithmeticHalfShare {
            <$inner as CircomGroth16Prover<P>>::to_half_share(a)
        }
        fn msm_public_points_hs<C>(points: &[Affine<C>], scalars: &[Self::ArithmeticHalfShare]) -> Self::PointHalfShare<Projective<C>>
        where
            C: SWCurveConfig<ScalarField = P::ScalarField>,
        {
            hip_msm::<P, C>(points, scalars) // all three protocols: half shares are plain field elements, the MSM is linear
        }
        fn scalar_mul_public_point_hs<C>(a: &C, b: Self::ArithmeticHalfShare) -> Self::PointHalfShare<C>
        where
            C: CurveGroup<ScalarField = P::ScalarField>,
        {
            <$inner as CircomGroth16Prover<P>>::scalar_mul_public_point_hs(a, b)
        }
        fn add_assign_points_public_hs<C: CurveGroup>(
            id: <Self::State as MpcState>::PartyID,
            a: &mut Self::PointHalfShare<C>,
            b: &C,
        ) {
            <$inner as CircomGroth16Prover<P>>::add_assign_points_public_hs(id, a, b)
        }
        fn open_half_point<N: Network, C>(a: Self::PointHalfShare<C>, net: &N, state: &mut Self::State) -> eyre::Result<C>
        where
            C: CurveGroup<ScalarField = P::ScalarField>,
        {
            <$inner as CircomGroth16Prover<P>>::open_half_point(a, net, state)
        }
        fn scalar_mul<N: Network>(
            a: &Self::PointHalfShare<P::G1>,
            b: Self::ArithmeticShare,
            net: &N,
            state: &mut Self::State,
        ) -> eyre::Result<Self::PointHalfShare<P::G1>> {
            <$inner as CircomGroth16Prover<P>>::scalar_mul(a, b, net, state) // incl. masking_ec_element (pointshare.rs:124)
        }
    };
}

/// Plain (single party) driver: mpc/plain.rs.
pub struct HipPlainGroth16Driver;
impl<P: Pairing> CircomGroth16Prover<P> for HipPlainGroth16Driver {
    type ArithmeticShare = P::ScalarField;
    type ArithmeticHalfShare = P::ScalarField;
    type PointHalfShare<C> = C where C: CurveGroup;
    type State = ();
    delegate_cold_methods!(PlainGroth16Driver);

    fn local_mul_vec(a: Vec<P::ScalarField>, b: Vec<P::ScalarField>, _: &mut ()) -> Vec<P::ScalarField> {
        let mut out = a; // in place: out == an input is allowed
        hip_ok(unsafe { sys::csh_vec_mul(curve_id::<P>(), limbs_of(&out), limbs_of(&b), limbs_mut(&mut out), b.len()) });
        out
    }
    fn distribute_powers_and_mul_by_const(coeffs: &mut [P::ScalarField], roots: &[P::ScalarField]) {
        assert_eq!(coeffs.len(), roots.len()); // zip_eq in the reference
        hip_ok(unsafe { sys::csh_vec_mul_table(curve_id::<P>(), limbs_mut(coeffs), limbs_of(roots), roots.len(), 1) });
    }
}

/// Replicated 3-party driver: mpc/rep3.rs.
pub struct HipRep3Groth16Driver;
impl<P: Pairing> CircomGroth16Prover<P> for HipRep3Groth16Driver {
    type ArithmeticShare = Rep3PrimeFieldShare<P::ScalarField>;
    type ArithmeticHalfShare = P::ScalarField;
    type PointHalfShare<C> = C where C: CurveGroup;
    type State = Rep3State;
    delegate_cold_methods!(Rep3Groth16Driver);

    /// rep3/arithmetic.rs:132-146: a.a*b.a + a.a*b.b + a.b*b.a + mask, the mask drawn exactly as the reference draws it
    /// (`masking_field_elements_vec`, rngs.rs:137-156) so the three parties' masks still cancel.
    fn local_mul_vec(a: Vec<Self::ArithmeticShare>, b: Vec<Self::ArithmeticShare>, state: &mut Rep3State) -> Vec<P::ScalarField> {
        assert_eq!(a.len(), b.len());
        match MASK_REQUEST.with(|m| m.get()) {
            // hip_reduction.rs::draw_mask: the mask of an n-entry product, operands empty (0 * 0 + mask without the zeros)
            MaskRequest::MaskOnly(n) if a.is_empty() => return state.rngs.rand.masking_field_elements_vec::<P::ScalarField>(n),
            // hip_reduction.rs::draw_seeds: one pair of fresh seeds through the public surface; all three parties consume the same
            // 2 x 32 words of their streams, so party i's first seed is party i+1's second and the device-made masks still cancel
            MaskRequest::Seeds if a.is_empty() => {
                SEEDS.with(|s| s.set(Some(state.rngs.rand.random_seeds())));
                return Vec::new();
            }
            _ => {}
        }
        let mask = state.rngs.rand.masking_field_elements_vec::<P::ScalarField>(a.len());
        let mut out = mask; // in place over the mask vector
        hip_ok(unsafe {
            sys::csh_rep3_local_mul_vec(curve_id::<P>(), limbs_of(&a), limbs_of(&b), limbs_of(&out), limbs_mut(&mut out), a.len())
        });
        out
    }
    /// mpc/rep3.rs:95-106: both components of every share times the public power
    fn distribute_powers_and_mul_by_const(coeffs: &mut [Self::ArithmeticShare], roots: &[P::ScalarField]) {
        assert_eq!(coeffs.len(), roots.len());
        hip_ok(unsafe { sys::csh_vec_mul_table(curve_id::<P>(), limbs_mut(coeffs), limbs_of(roots), roots.len(), 2) });
    }
}

/// Shamir driver: mpc/shamir.rs.
pub struct HipShamirGroth16Driver;
impl<P: Pairing> CircomGroth16Prover<P> for HipShamirGroth16Driver {
    type ArithmeticShare = ShamirPrimeFieldShare<P::ScalarField>;
    type ArithmeticHalfShare = P::ScalarField;
    type PointHalfShare<C> = C where C: CurveGroup;
    type State = ShamirState<P::ScalarField>;
    delegate_cold_methods!(ShamirGroth16Driver);

    /// shamir/arithmetic.rs:73-79: element-wise product of the shares (a degree-2t sharing)
    fn local_mul_vec(a: Vec<Self::ArithmeticShare>, b: Vec<Self::ArithmeticShare>, _: &mut Self::State) -> Vec<P::ScalarField> {
        assert_eq!(a.len(), b.len());
        let mut out = vec![<P::ScalarField as Default>::default(); a.len()];
        hip_ok(unsafe { sys::csh_vec_mul(curve_id::<P>(), limbs_of(&a), limbs_of(&b), limbs_mut(&mut out), a.len()) });
        out
    }
    fn distribute_powers_and_mul_by_const(coeffs: &mut [Self::ArithmeticShare], roots: &[P::ScalarField]) {
        assert_eq!(coeffs.len(), roots.len());
        hip_ok(unsafe { sys::csh_vec_mul_table(curve_id::<P>(), limbs_mut(coeffs), limbs_of(roots), roots.len(), 1) });
    }
}
