//! `CircomPlonkProver<P>` implementors whose four hot methods run on the GPU. Associated types = the reference drivers'
//! (co-plonk/src/mpc/{plain.rs:18-29, rep3.rs:17-26, shamir.rs:19-27}).
use crate::cold::plonk_cold_methods;
use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use ark_poly::EvaluationDomain;
use co_groth16_hip::bases;
use co_groth16_hip::domain::HipDomain;
use co_groth16_hip::error::hip_ok;
use co_groth16_hip::layout::{curve_id, limbs_mut, limbs_of, ncomp};
use co_plonk::mpc::{CircomPlonkProver, PlainPlonkDriver, Rep3PlonkDriver, ShamirPlonkDriver};
use core::mem::MaybeUninit;
use cosnarks_hip_sys as sys;
use mpc_core::protocols::rep3::{Rep3PointShare, Rep3PrimeFieldShare, Rep3State};
use mpc_core::protocols::shamir::{ShamirPointShare, ShamirPrimeFieldShare, ShamirState};
use mpc_core::MpcState;
use mpc_net::Network;

/// `EvaluationDomain::{fft, ifft}` on a vector of shares (`DomainCoeff` is component-wise, so a share of k field elements is k
/// interleaved transforms: `ncomp`): natural order in and out, the input zero-padded (or cut) to the domain size, `ifft` scaled by
/// 1/n -- ark-poly's semantics for a domain without coset offset, which is what co-plonk builds (`Domains::new`,
/// co-plonk/src/types.rs:70-109: Radix2EvaluationDomain with `group_gen` overwritten by the snarkjs root).
fn hip_transform<P: Pairing, S: Copy + Default, D: EvaluationDomain<P::ScalarField>>(data: &[S], domain: &D, inverse: bool) -> Vec<S> {
    use ark_ff::One;
    assert!(domain.coset_offset().is_one(), "cosnarks_hip: coset domains are not offloaded (co-plonk never builds one)");
    let n = domain.size();
    let gen = domain.group_gen();
    let dom = HipDomain::cached(curve_id::<P>(), domain.log_size_of_group() as u32, Some(&gen)).expect("domain");
    let mut v: Vec<S> = Vec::with_capacity(n);
    v.extend_from_slice(&data[..data.len().min(n)]);
    v.resize(n, S::default()); // fft_in_place: coeffs.resize(self.size(), T::zero())
    hip_ok(unsafe {
        if inverse {
            sys::csh_ifft(dom.raw(), limbs_mut(&mut v), ncomp::<S>())
        } else {
            sys::csh_fft(dom.raw(), limbs_mut(&mut v), ncomp::<S>())
        }
    });
    v
}

/// One MSM per share component over the same (cached, device-resident) bases: `ncomp::<S>()` results, Jacobian {x, y, z}.
fn hip_msm_shares<P: Pairing, C, S, const K: usize>(points: &[Affine<C>], scalars: &[S]) -> [Projective<C>; K]
where
    C: SWCurveConfig<ScalarField = P::ScalarField>,
{
    assert_eq!(ncomp::<S>() as usize, K);
    let n = points.len().min(scalars.len()); // msm_unchecked: the shorter of the two slices
    let (b, off) = bases::get_or_upload::<P, C>(points);
    let mut out: [MaybeUninit<Projective<C>>; K] = [const { MaybeUninit::uninit() }; K];
    let mut ptrs = [core::ptr::null_mut::<core::ffi::c_void>(); K];
    for (p, o) in ptrs.iter_mut().zip(out.iter_mut()) {
        *p = o.as_mut_ptr().cast();
    }
    hip_ok(unsafe { sys::csh_msm_shares(b.handle, off, n, limbs_of(scalars), K as u32, 1, ptrs.as_ptr()) });
    out.map(|o| unsafe { o.assume_init() }) // the library wrote (X, Y, Z), Z in {0, 1}
}

/// Plain (single party) driver: co-plonk/src/mpc/plain.rs.
pub struct HipPlainPlonkDriver;
impl<P, C> CircomPlonkProver<P> for HipPlainPlonkDriver
where
    P: Pairing<G1 = Projective<C>, G1Affine = Affine<C>>,
    C: SWCurveConfig<ScalarField = P::ScalarField>,
{
    type ArithmeticShare = P::ScalarField;
    type PointShareG1 = P::G1;
    type PointShareG2 = P::G2;
    type State = ();
    plonk_cold_methods!(PlainPlonkDriver);

    fn local_mul_vec(a: &[Self::ArithmeticShare], b: &[Self::ArithmeticShare], _state: &mut Self::State) -> Vec<P::ScalarField> {
        assert_eq!(a.len(), b.len());
        let mut out: Vec<P::ScalarField> = Vec::with_capacity(a.len());
        hip_ok(unsafe { sys::csh_vec_mul(curve_id::<P>(), limbs_of(a), limbs_of(b), out.as_mut_ptr().cast(), a.len()) });
        unsafe { out.set_len(a.len()) }; // SAFETY: CSH_OK = every element written
        out
    }
    fn fft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        hip_transform::<P, _, D>(data, domain, false)
    }
    fn ifft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        hip_transform::<P, _, D>(data, domain, true)
    }
    fn msm_public_points_g1(points: &[P::G1Affine], scalars: &[Self::ArithmeticShare]) -> Self::PointShareG1 {
        let [r] = hip_msm_shares::<P, C, _, 1>(points, scalars);
        r
    }
}

/// Replicated 3-party driver: co-plonk/src/mpc/rep3.rs.
pub struct HipRep3PlonkDriver;
impl<P, C> CircomPlonkProver<P> for HipRep3PlonkDriver
where
    P: Pairing<G1 = Projective<C>, G1Affine = Affine<C>>,
    C: SWCurveConfig<ScalarField = P::ScalarField>,
{
    type ArithmeticShare = Rep3PrimeFieldShare<P::ScalarField>;
    type PointShareG1 = Rep3PointShare<P::G1>;
    type PointShareG2 = Rep3PointShare<P::G2>;
    type State = Rep3State;
    plonk_cold_methods!(Rep3PlonkDriver);

    /// arithmetic::local_mul_vec (rep3/arithmetic.rs:132-146): lhs * rhs + mask per entry, the mask vector drawn exactly as the
    /// reference draws it (`masking_field_elements_vec`, rngs.rs:137-156), so the three parties' masks still cancel.
    fn local_mul_vec(a: &[Self::ArithmeticShare], b: &[Self::ArithmeticShare], state: &mut Self::State) -> Vec<P::ScalarField> {
        assert_eq!(a.len(), b.len());
        let mut out = state.rngs.rand.masking_field_elements_vec::<P::ScalarField>(a.len()); // in place over the mask vector
        hip_ok(unsafe { sys::csh_rep3_local_mul_vec(curve_id::<P>(), limbs_of(a), limbs_of(b), limbs_of(&out), limbs_mut(&mut out), a.len()) });
        out
    }
    fn fft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        hip_transform::<P, _, D>(data, domain, false) // both components through the same transform (rep3.rs:140-145)
    }
    fn ifft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        hip_transform::<P, _, D>(data, domain, true)
    }
    /// pointshare::msm_public_points (rep3/pointshare.rs:201-222): one MSM over the `a` components, one over the `b` components.
    fn msm_public_points_g1(points: &[P::G1Affine], scalars: &[Self::ArithmeticShare]) -> Self::PointShareG1 {
        let [a, b] = hip_msm_shares::<P, C, _, 2>(points, scalars);
        Rep3PointShare::new(a, b)
    }
}

/// Shamir driver: co-plonk/src/mpc/shamir.rs.
pub struct HipShamirPlonkDriver;
impl<P, C> CircomPlonkProver<P> for HipShamirPlonkDriver
where
    P: Pairing<G1 = Projective<C>, G1Affine = Affine<C>>,
    C: SWCurveConfig<ScalarField = P::ScalarField>,
{
    type ArithmeticShare = ShamirPrimeFieldShare<P::ScalarField>;
    type PointShareG1 = ShamirPointShare<P::G1>;
    type PointShareG2 = ShamirPointShare<P::G2>;
    type State = ShamirState<P::ScalarField>;
    plonk_cold_methods!(ShamirPlonkDriver);

    /// shamir/arithmetic.rs:73-79: element-wise product of the shares (a degree-2t sharing)
    fn local_mul_vec(a: &[Self::ArithmeticShare], b: &[Self::ArithmeticShare], _state: &mut Self::State) -> Vec<P::ScalarField> {
        assert_eq!(a.len(), b.len());
        let mut out: Vec<P::ScalarField> = Vec::with_capacity(a.len());
        hip_ok(unsafe { sys::csh_vec_mul(curve_id::<P>(), limbs_of(a), limbs_of(b), out.as_mut_ptr().cast(), a.len()) });
        unsafe { out.set_len(a.len()) };
        out
    }
    fn fft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        hip_transform::<P, _, D>(data, domain, false)
    }
    fn ifft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        hip_transform::<P, _, D>(data, domain, true)
    }
    /// shamir/pointshare.rs: the MSM of the share values (ShamirPrimeFieldShare is repr(transparent): one field element per entry)
    fn msm_public_points_g1(points: &[P::G1Affine], scalars: &[Self::ArithmeticShare]) -> Self::PointShareG1 {
        let [r] = hip_msm_shares::<P, C, _, 1>(points, scalars);
        ShamirPointShare::new(r)
    }
}
