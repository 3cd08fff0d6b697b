//! `R1CSToQAP` on the GPU (co-circom/co-groth16/src/groth16/reduction.rs:27-36). The FFT calls of the reference live INSIDE
//! its witness maps (reduction.rs:141-174, 255-328), not behind the driver trait, so the NTT offload is a new implementor of
//! this trait; it is a type parameter of every `prove`, hence a one-identifier change at the call site.
//!
//! ONE call into the library per witness map: the constraint matrices are resident on the device (`matrices.rs`, uploaded once
//! per circuit and GPU), the witness shares go up as they lie in memory, the whole of reduction.rs:99-192 (rows, six transforms,
//! two local multiplications, three coset shifts, the subtraction) runs on one stream, and `h` comes back into a vector that is
//! allocated but never zero-filled (the library populates its pages while the device works).
//!
//! Generic over `T: CircomGroth16Prover<P>` exactly like the reference, through the PUBLIC surface only:
//! * the protocol is read off the share type (one field element per entry = plain / Shamir, two = Rep3);
//! * the party index is read off `T::promote_to_trivial_shares(id, [1])` (rep3/arithmetic.rs `promote_to_trivial_share`:
//!   party 0 -> (1, 0), party 1 -> (0, 1), party 2 -> (0, 0)); `PartyID` itself carries no integer conversion in its bounds
//!   (mpc-core/src/lib.rs:21-31);
//! * the two Rep3 mask vectors are drawn by `T::local_mul_vec` on two zero vectors, which returns exactly the mask
//!   (rep3/arithmetic.rs:132-146), in the reference's order (reduction.rs:160 then :182): the party's generators advance as
//!   they would have, and the three parties' masks still cancel. The Hip Rep3 driver takes the length from a thread-local request
//!   and EMPTY operands (`drivers::MaskRequest::MaskOnly`): no zero vectors of n shares, no GPU call, just
//!   `masking_field_elements_vec(n)` (rngs.rs:137-156); a foreign `T` falls back to the zero-vector form;
//! * opt-in (`COSNARKS_HIP_SEEDED_MASKS=1`): ONE `Rep3Rand::random_seeds()` draw (rngs.rs:233, public) per witness map and both
//!   mask vectors generated on the device -- for sessions whose three parties all run this crate (`draw_seeds`).
use crate::domain::HipDomain;
use crate::error::check;
use crate::layout::{curve_id, limbs, limbs_of, ncomp};
use crate::{drivers, matrices};
use ark_ec::pairing::Pairing;
use ark_ff::{FftField, Field, LegendreSymbol, One, PrimeField, Zero};
use co_groth16::mpc::CircomGroth16Prover;
use co_groth16::R1CSToQAP;
use core::mem::size_of;
use cosnarks_hip_sys as sys;
use mpc_core::MpcState;
use taceo_groth16::ConstraintMatrices;

/// snarkjs' roots of unity, restated: `roots_of_unity` / `groth16_roots_of_unity` are PRIVATE functions of the reference
/// (co-circom/co-groth16/src/groth16.rs:60-74 and :91-100; `lib.rs:9-13` exports neither), so an out-of-tree implementor of
/// `R1CSToQAP` has to carry its own copy. q = the smallest quadratic non-residue, z = q^TRACE generates the 2^TWO_ADICITY-th
/// roots; `roots[i]` generates the domain of size 2^i.
fn roots_of_unity<F: PrimeField + FftField>() -> (F, Vec<F>) {
    let mut roots = vec![F::zero(); F::TWO_ADICITY as usize + 1];
    let mut q = F::one();
    while q.legendre() != LegendreSymbol::QuadraticNonResidue {
        q += F::one();
    }
    roots[0] = q.pow(F::TRACE);
    for i in 1..roots.len() {
        roots[i] = roots[i - 1].square();
    }
    roots.reverse();
    (q, roots)
}

/// (generator of the domain of size 2^pow, shift onto the odd coset = generator of the domain twice as large); for
/// pow == TWO_ADICITY there is no larger domain and snarkjs takes q^2 (groth16.rs:91-100).
fn groth16_roots_of_unity<F: PrimeField + FftField>(pow: usize) -> (F, F) {
    let (q, roots) = roots_of_unity::<F>();
    let group_gen = roots[pow];
    let coset_shift = if F::TWO_ADICITY as usize == pow { q.square() } else { roots[pow + 1] };
    (group_gen, coset_shift)
}

/// What the library needs to know about the driver, learnt through its public methods.
struct Protocol {
    id: i32,    // 0 = one field element per share (plain, Shamir), 1 = Rep3 {a, b}
    party: i32, // Rep3: where public terms go (0 -> a, 1 -> b, 2 -> nowhere); 0 otherwise
}

fn protocol_of<P: Pairing, T: CircomGroth16Prover<P>>(id: <T::State as MpcState>::PartyID) -> eyre::Result<Protocol> {
    assert_eq!(size_of::<T::ArithmeticHalfShare>(), size_of::<P::ScalarField>(), "half shares are single field elements (mpc.rs:36-49)");
    match ncomp::<T::ArithmeticShare>() {
        1 => Ok(Protocol { id: 0, party: 0 }),
        2 => {
            let one = T::promote_to_trivial_shares(id, &[P::ScalarField::one()]);
            // SAFETY: a two-component share is {a, b}: two consecutive field elements (layout.rs asserts it for Rep3PrimeFieldShare)
            let ab: [P::ScalarField; 2] = unsafe { core::ptr::read((one.as_ptr()).cast()) };
            let party = match (ab[0].is_one(), ab[1].is_one(), ab[0].is_zero(), ab[1].is_zero()) {
                (true, _, _, true) => 0,
                (_, true, true, _) => 1,
                (_, _, true, true) => 2,
                _ => eyre::bail!("unrecognised two-component share type: promote_to_trivial_shares(1) is neither (1,0), (0,1) nor (0,0)"),
            };
            Ok(Protocol { id: 1, party })
        }
        k => eyre::bail!("unsupported share type: {k} field elements per entry"),
    }
}

/// The mask vector the next `T::local_mul_vec` of `n` entries would add. On the Hip Rep3 driver the length travels in a thread-local
/// and the operands are EMPTY (`drivers::MaskRequest::MaskOnly`): nothing of size n is built but the mask itself. Any other
/// `T` ignores the request and returns the (empty) product of the empty operands without touching its generators -- then, and only
/// then, the generic form runs: the product of two zero vectors of n shares is the mask (rep3/arithmetic.rs:132-146).
fn draw_mask<P: Pairing, T: CircomGroth16Prover<P>>(state: &mut T::State, n: usize) -> Vec<T::ArithmeticHalfShare> {
    let m = drivers::with_mask_request(drivers::MaskRequest::MaskOnly(n), || T::local_mul_vec(Vec::new(), Vec::new(), state));
    if m.len() == n {
        return m;
    }
    debug_assert!(m.is_empty());
    let zeros = || vec![T::ArithmeticShare::default(); n];
    T::local_mul_vec(zeros(), zeros(), state)
}

/// Opt-in all-GPU-parties mode (`COSNARKS_HIP_SEEDED_MASKS=1`, read once): one pair of fresh correlated seeds per witness map through
/// the PUBLIC `Rep3Rand::random_seeds()` (rngs.rs:233) instead of two host mask vectors (2 x 2 n `from_be_bytes_mod_order` + 2 x 32 n
/// keystream bytes on the host at every witness map); the device then generates both mask vectors with its ChaCha12
/// (`csh_groth16_witness_map`: chunks [0, n) of both streams for "c", [n, 2n) for "ab"). Every party advances its two generators by
/// the same 32 words, party i's first seed equals party i+1's second, so the masks of the three parties still sum to zero -- but they
/// are NOT the values a reference CPU party would draw: all three parties of a session must run this mode. None: `T` is not the Hip
/// Rep3 driver (the request was ignored) -- the caller falls back to host masks.
fn draw_seeds<P: Pairing, T: CircomGroth16Prover<P>>(state: &mut T::State) -> Option<([u8; 32], [u8; 32])> {
    let _ = drivers::take_seeds(); // nothing stale
    let r = drivers::with_mask_request(drivers::MaskRequest::Seeds, || T::local_mul_vec(Vec::new(), Vec::new(), state));
    debug_assert!(r.is_empty());
    drivers::take_seeds()
}

fn seeded_masks_enabled() -> bool {
    static ON: std::sync::OnceLock<bool> = std::sync::OnceLock::new();
    *ON.get_or_init(|| std::env::var("COSNARKS_HIP_SEEDED_MASKS").map(|v| v != "0" && !v.is_empty()).unwrap_or(false))
}

/// `n` half shares of allocated, uninitialised memory for the library to fill (no zero fill: 4.7 ms per 32 MB on the GPU hosts).
fn uninit_half_shares<H>(n: usize) -> Vec<H> {
    Vec::with_capacity(n)
}

/// snarkjs' witness map (the odd coset of a domain twice as large): drop-in for `CircomReduction` (reduction.rs:62-193).
pub struct HipCircomReduction;

impl R1CSToQAP for HipCircomReduction {
    fn witness_map_from_matrices<P: Pairing, T: CircomGroth16Prover<P>>(
        state: &mut T::State,
        matrices: &ConstraintMatrices<P::ScalarField>,
        public_inputs: &[P::ScalarField],
        private_witness: &[T::ArithmeticShare],
    ) -> eyre::Result<Vec<T::ArithmeticHalfShare>> {
        let num_constraints = matrices.num_constraints;
        let num_inputs = matrices.num_instance_variables;
        let domain_size = (num_constraints + num_inputs).next_power_of_two(); // reduction.rs:84-86
        let power = domain_size.ilog2() as usize;
        if power > <P::ScalarField as FftField>::TWO_ADICITY as usize {
            eyre::bail!("Polynomial Degree too large"); // reduction.rs:87-89
        }
        // snarkjs' root for the domain and the root of the domain twice as large as the coset shift (:90-94)
        let (group_gen, coset_shift) = groth16_roots_of_unity::<P::ScalarField>(power);
        let dom = HipDomain::cached(curve_id::<P>(), power as u32, Some(&group_gen))?; // Domain::with_group_gen (:93)
        let proto = protocol_of::<P, T>(state.id())?;
        let dm = matrices::get_or_upload::<P>(matrices, false)?;
        let mut h = uninit_half_shares::<T::ArithmeticHalfShare>(domain_size);
        if proto.id == 1 && seeded_masks_enabled() {
            if let Some((seed1, seed2)) = draw_seeds::<P, T>(state) {
                check(unsafe {
                    sys::csh_groth16_witness_map(dom.raw, limbs(&coset_shift), proto.id, proto.party, dm.a.handle, dm.b.handle, num_constraints,
                                                 limbs_of(public_inputs), num_inputs.min(public_inputs.len()), limbs_of(private_witness),
                                                 private_witness.len(), seed1.as_ptr(), 0, seed2.as_ptr(), 0, h.as_mut_ptr().cast())
                })?;
                unsafe { h.set_len(domain_size) }; // SAFETY: CSH_OK: all `domain_size` elements were written
                return Ok(h);
            }
        }
        // the two mask vectors in the reference's order: "c: local_mul_vec" (:160), then "ab" (:182)
        let (mask_c, mask_ab) = if proto.id == 1 {
            (draw_mask::<P, T>(state, domain_size), draw_mask::<P, T>(state, domain_size))
        } else {
            (Vec::new(), Vec::new())
        };
        let mask_ptr = |m: &Vec<T::ArithmeticHalfShare>| if m.is_empty() { core::ptr::null() } else { limbs_of(m) };
        check(unsafe {
            sys::csh_groth16_witness_map_masks(dom.raw, limbs(&coset_shift), proto.id, proto.party, dm.a.handle, dm.b.handle, num_constraints,
                                               limbs_of(public_inputs), num_inputs.min(public_inputs.len()), limbs_of(private_witness),
                                               private_witness.len(), mask_ptr(&mask_c), mask_ptr(&mask_ab), h.as_mut_ptr().cast())
        })?;
        // SAFETY: the call returned CSH_OK, so all `domain_size` elements were written (canonical Montgomery limbs = valid field elements)
        unsafe { h.set_len(domain_size) };
        Ok(h)
    }
}

/// arkworks' LibSnark witness map, (A B - C) / Z through coset evaluations: drop-in for `LibSnarkReduction` (reduction.rs:237-342).
pub struct HipLibSnarkReduction;

impl R1CSToQAP for HipLibSnarkReduction {
    fn witness_map_from_matrices<P: Pairing, T: CircomGroth16Prover<P>>(
        state: &mut T::State,
        matrices: &ConstraintMatrices<P::ScalarField>,
        public_inputs: &[P::ScalarField],
        private_witness: &[T::ArithmeticShare],
    ) -> eyre::Result<Vec<T::ArithmeticHalfShare>> {
        let num_constraints = matrices.num_constraints;
        let num_inputs = matrices.num_instance_variables;
        let domain_size = (num_constraints + num_inputs).next_power_of_two(); // Domain::new(n) (:249)
        let log_n = domain_size.trailing_zeros();
        if log_n > <P::ScalarField as FftField>::TWO_ADICITY {
            eyre::bail!("Polynomial Degree too large");
        }
        let dom = HipDomain::cached::<P::ScalarField>(curve_id::<P>(), log_n, None)?;
        let proto = protocol_of::<P, T>(state.id())?;
        let dm = matrices::get_or_upload::<P>(matrices, true)?;
        let g = <P::ScalarField as FftField>::GENERATOR; // the coset g * <w> (:255)
        // one local_mul_vec (:289), one mask vector
        let mask = if proto.id == 1 { draw_mask::<P, T>(state, domain_size) } else { Vec::new() };
        let mut h = uninit_half_shares::<T::ArithmeticHalfShare>(domain_size);
        check(unsafe {
            sys::csh_groth16_witness_map_libsnark_masks(dom.raw, limbs(&g), proto.id, proto.party, dm.a.handle, dm.b.handle,
                                                        dm.c.as_ref().expect("uploaded with the C side").handle, num_constraints,
                                                        limbs_of(public_inputs), num_inputs.min(public_inputs.len()), limbs_of(private_witness),
                                                        private_witness.len(), if mask.is_empty() { core::ptr::null() } else { limbs_of(&mask) },
                                                        h.as_mut_ptr().cast())
        })?;
        unsafe { h.set_len(domain_size) }; // SAFETY: as above
        Ok(h)
    }
}
