//! In-memory layouts the C ABI relies on. The reference passes `&[F]`, `&[Affine<C>]`, `&[Rep3PrimeFieldShare<F>]`
//! straight through; these assertions fail the build if arkworks / mpc-core ever change them.
use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, Projective, SWCurveConfig};
use core::ffi::c_int;
use core::mem::{align_of, size_of};
use mpc_core::protocols::rep3::Rep3PrimeFieldShare;
use mpc_core::protocols::shamir::ShamirPrimeFieldShare;

// Fp<MontBackend<_, N>, N> is exactly its N little-endian u64 Montgomery limbs.
const _: () = assert!(size_of::<ark_bn254::Fr>() == 32 && align_of::<ark_bn254::Fr>() == 8);
const _: () = assert!(size_of::<ark_bn254::Fq>() == 32);
const _: () = assert!(size_of::<ark_bls12_381::Fr>() == 32);
const _: () = assert!(size_of::<ark_bls12_381::Fq>() == 48);
// Rep3 share = {a, b}, two consecutive field elements (mpc-core/src/protocols/rep3/arithmetic/types.rs:21-28; not repr(C),
// hence this assertion); Shamir share is repr(transparent) (shamir/arithmetic/types.rs:9-13).
const _: () = assert!(size_of::<Rep3PrimeFieldShare<ark_bn254::Fr>>() == 64);
const _: () = assert!(size_of::<ShamirPrimeFieldShare<ark_bn254::Fr>>() == 32);
// Affine<C> = {x, y, infinity: bool} + padding: passed with stride_bytes = size_of::<Affine<C>>() (72 on BN254 G1); the flag byte
// is ignored by the library, so infinity must be the all-zero encoding (what the zkey parser produces, see `bases.rs`).
const _: () = assert!(size_of::<Affine<ark_bn254::g1::Config>>() == 72);
const _: () = assert!(size_of::<Affine<ark_bn254::g2::Config>>() == 136);
const _: () = assert!(size_of::<Affine<ark_bls12_381::g1::Config>>() == 104);
const _: () = assert!(size_of::<Affine<ark_bls12_381::g2::Config>>() == 200);
// Projective<C> = Jacobian {x, y, z}: the library writes (X, Y, Z) with Z in {0, 1}; infinity = (1, 1, 0).
const _: () = assert!(size_of::<Projective<ark_bn254::g1::Config>>() == 96);
const _: () = assert!(size_of::<Projective<ark_bls12_381::g2::Config>>() == 288);

/// Pairing -> `csh_curve_t`.
pub trait HipCurve: Pairing {
    const CURVE: c_int;
}
impl HipCurve for ark_bn254::Bn254 {
    const CURVE: c_int = cosnarks_hip_sys::CSH_BN254;
}
impl HipCurve for ark_bls12_381::Bls12_381 {
    const CURVE: c_int = cosnarks_hip_sys::CSH_BLS12_381;
}

/// Curve id of a pairing known only through the generic parameter of the reference's traits (they are bounded by `Pairing`,
/// not by [`HipCurve`]): decided by the scalar-field modulus, once.
pub fn curve_id<P: Pairing>() -> c_int {
    use ark_ff::PrimeField;
    let bits = <P::ScalarField as PrimeField>::MODULUS_BIT_SIZE;
    let base_bits = <P::BaseField as PrimeField>::MODULUS_BIT_SIZE;
    match (bits, base_bits) {
        (254, 254) => cosnarks_hip_sys::CSH_BN254,
        (255, 381) => cosnarks_hip_sys::CSH_BLS12_381,
        (253, 377) => cosnarks_hip_sys::CSH_BLS12_377, // the LibSnarkReduction fixtures' curve (co-groth16/src/lib.rs:231-300): MSM + NTT + share vectors
        _ => panic!("cosnarks_hip: unsupported pairing ({bits}-bit scalar field, {base_bits}-bit base field)"),
    }
}

/// G1 or G2 of the curve, decided by the coordinate width (`C::BaseField` is Fq or Fq2).
pub fn group_id<C: SWCurveConfig>() -> c_int {
    use ark_ff::Field;
    if <C::BaseField as Field>::extension_degree() == 1 { cosnarks_hip_sys::CSH_G1 } else { cosnarks_hip_sys::CSH_G2 }
}

#[inline]
pub fn limbs<T>(x: &T) -> *const u64 {
    (x as *const T).cast()
}
#[inline]
pub fn limbs_of<T>(xs: &[T]) -> *const u64 {
    xs.as_ptr().cast()
}
#[inline]
pub fn limbs_mut<T>(xs: &mut [T]) -> *mut u64 {
    xs.as_mut_ptr().cast()
}
/// 32-byte field elements per share entry: 1 for `F` / Shamir shares, 2 for Rep3 shares (DomainCoeff<F>).
#[inline]
pub const fn ncomp<S>() -> u32 {
    (size_of::<S>() / 32) as u32
}
