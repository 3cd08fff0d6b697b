//! `NoirUltraHonkProver<P>` implementors whose four hot methods run on the GPU, for `P` = BN254 G1 or Grumpkin (the two
//! `HonkCurve<TranscriptFieldType>` implementors, honk_curve.rs:47-191). Associated types = the reference drivers'
//! (co-noir-common/src/mpc/{plain.rs:27-30, rep3.rs:24-31, shamir.rs:21-24}).
use crate::cold::honk_cold_methods;
use ark_ec::CurveGroup;
use ark_ff::PrimeField;
use ark_poly::EvaluationDomain;
use co_groth16_hip::bases;
use co_groth16_hip::domain::HipDomain;
use co_groth16_hip::error::hip_ok;
use co_groth16_hip::layout::{limbs_mut, limbs_of, ncomp};
use co_noir_common::honk_proof::TranscriptFieldType;
use co_noir_common::mpc::plain::PlainUltraHonkDriver;
use co_noir_common::mpc::rep3::Rep3UltraHonkDriver;
use co_noir_common::mpc::shamir::ShamirUltraHonkDriver;
use co_noir_common::mpc::NoirUltraHonkProver;
use co_noir_common::honk_curve::HonkCurve;
use core::any::TypeId;
use core::mem::{size_of, MaybeUninit};
use cosnarks_hip_sys as sys;
use mpc_core::protocols::rep3::{Rep3PointShare, Rep3PrimeFieldShare, Rep3State};
use mpc_core::protocols::shamir::{ShamirPointShare, ShamirPrimeFieldShare, ShamirState};
use mpc_core::uint::FieldUint;
use mpc_core::{MpcState, PlainState};
use mpc_net::Network;

/// `csh_curve_t` of a Honk curve: BN254 G1 (scalars in Fr) or Grumpkin (the cycle curve: scalars in BN254 Fq, coordinates in Fr).
fn honk_curve_id<P: CurveGroup>() -> i32 {
    if TypeId::of::<P>() == TypeId::of::<ark_bn254::G1Projective>() {
        sys::CSH_BN254
    } else if TypeId::of::<P>() == TypeId::of::<ark_grumpkin::Projective>() {
        sys::CSH_GRUMPKIN
    } else {
        panic!("cosnarks_hip: unsupported Honk curve")
    }
}
/// Scalar-field entry points (vector products, transforms) take the curve whose Fr the elements live in: BN254 for BN254 G1;
/// Grumpkin's scalar field is BN254 Fq, which the library's NTT / vector kernels do not carry -- those methods delegate.
fn scalar_field_id<P: CurveGroup>() -> Option<i32> {
    (TypeId::of::<P>() == TypeId::of::<ark_bn254::G1Projective>()).then_some(sys::CSH_BN254)
}

/// One MSM per share component (`K` = field elements per share entry) over the cached, device-resident bases: the library writes
/// Jacobian (X, Y, Z) with Z in {0, 1}, which is `Projective<C>`'s layout for both Honk curves (co-groth16-hip's layout.rs asserts it
/// for the BN254 instance; Grumpkin's `Projective` is the same struct over the swapped fields).
fn msm_components<P: CurveGroup, S, const K: usize>(points: &[P::Affine], scalars: &[S]) -> [P; K] {
    assert_eq!(ncomp::<S>() as usize, K);
    assert_eq!(size_of::<P>(), 3 * size_of::<P::BaseField>(), "Projective {{x, y, z}}");
    let n = points.len().min(scalars.len()); // msm_unchecked: the shorter of the two slices (honk_curve.rs:33-35)
    // SAFETY: `points` is a live slice for the duration of the call
    let (b, off) = unsafe { bases::get_or_upload_raw(honk_curve_id::<P>(), sys::CSH_G1, points.as_ptr().cast(), points.len(), size_of::<P::Affine>(), points.len()) };
    let mut out: [MaybeUninit<P>; K] = [const { MaybeUninit::uninit() }; K];
    let mut ptrs = [core::ptr::null_mut::<core::ffi::c_void>(); K];
    for (p, o) in ptrs.iter_mut().zip(out.iter_mut()) {
        *p = o.as_mut_ptr().cast();
    }
    hip_ok(unsafe { sys::csh_msm_shares(b.handle, off, n, limbs_of(scalars), K as u32, 1, ptrs.as_ptr()) });
    out.map(|o| unsafe { o.assume_init() })
}

/// `HonkCurve::fast_msm` on the GPU (honk_curve.rs:35): `msm_unchecked(bases, scalars)` for BN254 G1 and Grumpkin.
pub fn hip_fast_msm<P: HonkCurve<TranscriptFieldType>>(bases: &[P::Affine], scalars: &[P::ScalarField]) -> P {
    let [r] = msm_components::<P, _, 1>(bases, scalars);
    r
}

/// `EvaluationDomain::{fft, ifft}` on a vector of shares, as in co-plonk-hip (natural order, zero-padded, 1/n on the way back).
fn transform<P: CurveGroup, S: Copy + Default, D: EvaluationDomain<P::ScalarField>>(field: i32, data: &[S], domain: &D, inverse: bool) -> Vec<S> {
    use ark_ff::One;
    assert!(domain.coset_offset().is_one(), "cosnarks_hip: coset domains are not offloaded");
    let n = domain.size();
    let gen = domain.group_gen();
    let dom = HipDomain::cached(field, domain.log_size_of_group() as u32, Some(&gen)).expect("domain");
    let mut v: Vec<S> = Vec::with_capacity(n);
    v.extend_from_slice(&data[..data.len().min(n)]);
    v.resize(n, S::default());
    hip_ok(unsafe { if inverse { sys::csh_ifft(dom.raw(), limbs_mut(&mut v), ncomp::<S>()) } else { sys::csh_fft(dom.raw(), limbs_mut(&mut v), ncomp::<S>()) } });
    v
}

fn vec_mul<F: PrimeField, S>(field: i32, a: &[S], b: &[S]) -> Vec<F> {
    assert_eq!(a.len(), b.len());
    assert_eq!(size_of::<S>(), size_of::<F>());
    let mut out: Vec<F> = Vec::with_capacity(a.len());
    hip_ok(unsafe { sys::csh_vec_mul(field, limbs_of(a), limbs_of(b), out.as_mut_ptr().cast(), a.len()) });
    unsafe { out.set_len(a.len()) }; // SAFETY: CSH_OK = every element written
    out
}

/// Plain driver: co-noir-common/src/mpc/plain.rs.
#[derive(Clone, Debug)]
pub struct HipPlainUltraHonkDriver;
impl<P: HonkCurve<TranscriptFieldType>> NoirUltraHonkProver<P> for HipPlainUltraHonkDriver {
    type ArithmeticShare = P::ScalarField;
    type PointShare = P;
    type State = PlainState;
    honk_cold_methods!(PlainUltraHonkDriver);

    fn local_mul_vec(a: &[Self::ArithmeticShare], b: &[Self::ArithmeticShare], state: &mut Self::State) -> Vec<P::ScalarField> {
        match scalar_field_id::<P>() {
            Some(f) => vec_mul::<P::ScalarField, _>(f, a, b),
            None => <PlainUltraHonkDriver as NoirUltraHonkProver<P>>::local_mul_vec(a, b, state),
        }
    }
    fn msm_public_points(points: &[P::Affine], scalars: &[Self::ArithmeticShare]) -> Self::PointShare {
        hip_fast_msm::<P>(points, scalars) // plain.rs:271: P::fast_msm(points, scalars)
    }
    fn fft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        match scalar_field_id::<P>() {
            Some(f) => transform::<P, _, D>(f, data, domain, false),
            None => <PlainUltraHonkDriver as NoirUltraHonkProver<P>>::fft::<D>(data, domain),
        }
    }
    fn ifft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        match scalar_field_id::<P>() {
            Some(f) => transform::<P, _, D>(f, data, domain, true),
            None => <PlainUltraHonkDriver as NoirUltraHonkProver<P>>::ifft::<D>(data, domain),
        }
    }
}

/// Replicated 3-party driver: co-noir-common/src/mpc/rep3.rs.
#[derive(Debug)]
pub struct HipRep3UltraHonkDriver;
impl<P: HonkCurve<TranscriptFieldType>> NoirUltraHonkProver<P> for HipRep3UltraHonkDriver
where
    P::ScalarField: FieldUint,
    P::BaseField: FieldUint<Uint = <P::ScalarField as FieldUint>::Uint>,
{
    type ArithmeticShare = Rep3PrimeFieldShare<P::ScalarField>;
    type PointShare = Rep3PointShare<P>;
    type State = Rep3State;
    honk_cold_methods!(Rep3UltraHonkDriver);

    /// arithmetic::local_mul_vec (rep3/arithmetic.rs:132-146), masks drawn as the reference draws them (rngs.rs:137-156)
    fn local_mul_vec(a: &[Self::ArithmeticShare], b: &[Self::ArithmeticShare], state: &mut Self::State) -> Vec<P::ScalarField> {
        let Some(f) = scalar_field_id::<P>() else {
            return <Rep3UltraHonkDriver as NoirUltraHonkProver<P>>::local_mul_vec(a, b, state);
        };
        assert_eq!(a.len(), b.len());
        let mut out = state.rngs.rand.masking_field_elements_vec::<P::ScalarField>(a.len()); // in place over the mask vector
        hip_ok(unsafe { sys::csh_rep3_local_mul_vec(f, limbs_of(a), limbs_of(b), limbs_of(&out), limbs_mut(&mut out), a.len()) });
        out
    }
    /// rep3.rs:259-266: the reference unzips the shares on the host and runs two `fast_msm`; here the {a, b} vector goes up once
    fn msm_public_points(points: &[P::Affine], scalars: &[Self::ArithmeticShare]) -> Self::PointShare {
        let [a, b] = msm_components::<P, _, 2>(points, scalars);
        Rep3PointShare::new(a, b)
    }
    fn fft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        match scalar_field_id::<P>() {
            Some(f) => transform::<P, _, D>(f, data, domain, false),
            None => <Rep3UltraHonkDriver as NoirUltraHonkProver<P>>::fft::<D>(data, domain),
        }
    }
    fn ifft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        match scalar_field_id::<P>() {
            Some(f) => transform::<P, _, D>(f, data, domain, true),
            None => <Rep3UltraHonkDriver as NoirUltraHonkProver<P>>::ifft::<D>(data, domain),
        }
    }
}

/// Shamir driver: co-noir-common/src/mpc/shamir.rs.
#[derive(Debug)]
pub struct HipShamirUltraHonkDriver;
impl<P: HonkCurve<TranscriptFieldType>> NoirUltraHonkProver<P> for HipShamirUltraHonkDriver {
    type ArithmeticShare = ShamirPrimeFieldShare<P::ScalarField>;
    type PointShare = ShamirPointShare<P>;
    type State = ShamirState<P::ScalarField>;
    honk_cold_methods!(ShamirUltraHonkDriver);

    /// shamir/arithmetic.rs:73-79 (ShamirPrimeFieldShare is repr(transparent): one field element per entry)
    fn local_mul_vec(a: &[Self::ArithmeticShare], b: &[Self::ArithmeticShare], state: &mut Self::State) -> Vec<P::ScalarField> {
        match scalar_field_id::<P>() {
            Some(f) => vec_mul::<P::ScalarField, _>(f, a, b),
            None => <ShamirUltraHonkDriver as NoirUltraHonkProver<P>>::local_mul_vec(a, b, state),
        }
    }
    /// shamir.rs:250-256: fast_msm over the share values
    fn msm_public_points(points: &[P::Affine], scalars: &[Self::ArithmeticShare]) -> Self::PointShare {
        let [r] = msm_components::<P, _, 1>(points, scalars);
        ShamirPointShare::new(r)
    }
    fn fft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        match scalar_field_id::<P>() {
            Some(f) => transform::<P, _, D>(f, data, domain, false),
            None => <ShamirUltraHonkDriver as NoirUltraHonkProver<P>>::fft::<D>(data, domain),
        }
    }
    fn ifft<D: EvaluationDomain<P::ScalarField>>(data: &[Self::ArithmeticShare], domain: &D) -> Vec<Self::ArithmeticShare> {
        match scalar_field_id::<P>() {
            Some(f) => transform::<P, _, D>(f, data, domain, true),
            None => <ShamirUltraHonkDriver as NoirUltraHonkProver<P>>::ifft::<D>(data, domain),
        }
    }
}
