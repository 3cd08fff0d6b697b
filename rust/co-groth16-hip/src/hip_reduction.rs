//! `R1CSToQAP` on the GPU (co-circom/co-groth16/src/groth16/reduction.rs:27-36). The FFT calls of the reference live INSIDE
//! its witness maps (reduction.rs:141-174, 255-328), not behind the driver trait, so the NTT offload is a new implementor of
//! this trait; it is a type parameter of every `prove`, hence a one-identifier change at the call site.
//!
//! The implementor is generic over `T: CircomGroth16Prover<P>` exactly like the reference: constraint rows are evaluated with
//! `T::evaluate_constraint` (host, rayon -- or on the device through `witness_map_device` below when the matrices are
//! resident), the transforms run through the C ABI on the share vectors as they lie in memory (`ncomp` = 1 or 2 field elements
//! per entry), and `T::local_mul_vec` / `T::distribute_powers_and_mul_by_const` are the driver's (GPU-backed for the Hip
//! drivers, the reference's otherwise -- both give the same field elements).
use crate::domain::HipDomain;
use crate::error::check;
use crate::layout::{curve_id, limbs, limbs_mut, limbs_of};
use ark_ec::pairing::Pairing;
use ark_ff::{FftField, Field, LegendreSymbol, One, PrimeField};
use ark_relations::utils::matrix::Matrix;
use co_groth16::mpc::CircomGroth16Prover;
use co_groth16::R1CSToQAP;
use cosnarks_hip_sys as sys;
use mpc_core::MpcState;
use rayon::prelude::*;
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

fn eval_rows<P: Pairing, T: CircomGroth16Prover<P>>(
    id: <T::State as MpcState>::PartyID,
    rows: &Matrix<P::ScalarField>,
    public_inputs: &[P::ScalarField],
    private_witness: &[T::ArithmeticShare],
    domain_size: usize,
) -> Vec<T::ArithmeticShare> {
    // reduction.rs:196-210: one sparse dot product per constraint, zero-padded to the domain
    let mut out: Vec<T::ArithmeticShare> =
        rows.par_iter().with_min_len(256).map(|r| T::evaluate_constraint(id, r, public_inputs, private_witness)).collect();
    out.resize(domain_size, T::ArithmeticShare::default());
    out
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
        let dom = HipDomain::new(curve_id::<P>(), power as u32, Some(&group_gen))?; // Domain::with_group_gen (:93)
        let id = state.id();

        // rows of A and B; the public inputs take the slots after the constraints (:99-113)
        let mut a = eval_rows::<P, T>(id, &matrices.a, public_inputs, private_witness, domain_size);
        let mut b = eval_rows::<P, T>(id, &matrices.b, public_inputs, private_witness, domain_size);
        let promoted = T::promote_to_trivial_shares(id, public_inputs);
        a[num_constraints..num_constraints + num_inputs].clone_from_slice(&promoted[..num_inputs]);

        // c = a * b in the evaluation domain, before a and b leave it (:135-160)
        let mut c = T::local_mul_vec(a.clone(), b.clone(), state);
        // a, b, c onto the odd coset: ifft -> multiply by shift^i (bit-reversed table) -> fft (:141-174)
        let table = dom.coset_table(&coset_shift);
        dom.ifft_in_to_out(&mut a);
        dom.ifft_in_to_out(&mut b);
        dom.ifft_in_to_out(&mut c);
        T::distribute_powers_and_mul_by_const(&mut a, &table);
        T::distribute_powers_and_mul_by_const(&mut b, &table);
        check(unsafe { sys::csh_vec_mul_table(curve_id::<P>(), limbs_mut(&mut c), limbs_of(&table), domain_size, 1) })?; // c *= table (:166-171)
        dom.fft_out_to_in(&mut a);
        dom.fft_out_to_in(&mut b);
        dom.fft_out_to_in(&mut c);
        // h = a * b - c on the coset (:176-192)
        let mut ab = T::local_mul_vec(a, b, state);
        check(unsafe { sys::csh_vec_sub(curve_id::<P>(), limbs_of(&ab), limbs_of(&c), limbs_mut(&mut ab), domain_size, 1) })?;
        Ok(ab)
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
        let dom = HipDomain::new::<P::ScalarField>(curve_id::<P>(), log_n, None)?;
        let id = state.id();
        let mut a = eval_rows::<P, T>(id, &matrices.a, public_inputs, private_witness, domain_size);
        let mut b = eval_rows::<P, T>(id, &matrices.b, public_inputs, private_witness, domain_size);
        let promoted = T::promote_to_trivial_shares(id, public_inputs);
        a[num_constraints..num_constraints + num_inputs].clone_from_slice(&promoted[..num_inputs]);
        let mut c: Vec<T::ArithmeticHalfShare> = matrices
            .c
            .par_iter()
            .with_min_len(256)
            .map(|r| T::evaluate_constraint_half_share(id, r, public_inputs, private_witness))
            .collect();
        c.resize(domain_size, T::ArithmeticHalfShare::default());
        // coefficients, then evaluations on the coset g * <w> (g = F::GENERATOR), (:255-300)
        let g = <P::ScalarField as FftField>::GENERATOR;
        let table = dom.coset_table(&g);
        for v in [&mut a, &mut b] {
            dom.ifft_in_to_out(v);
            T::distribute_powers_and_mul_by_const(v, &table);
            dom.fft_out_to_in(v);
        }
        dom.ifft_in_to_out(&mut c);
        check(unsafe { sys::csh_vec_mul_table(curve_id::<P>(), limbs_mut(&mut c), limbs_of(&table), domain_size, 1) })?;
        dom.fft_out_to_in(&mut c);
        let mut ab = T::local_mul_vec(a, b, state);
        check(unsafe { sys::csh_vec_sub(curve_id::<P>(), limbs_of(&ab), limbs_of(&c), limbs_mut(&mut ab), domain_size, 1) })?;
        // divide by Z(g x) = g^n - 1 (constant on the coset), interpolate, undo the shift (:312-340): the inverse powers are applied
        // in bit-reversed order straight after ifft_in_to_out, then the permutation -- the same values as permuting first
        let z_inv = (g.pow([domain_size as u64]) - P::ScalarField::one()).inverse().expect("g^n != 1");
        let g_inv = g.inverse().expect("generator is non-zero");
        let mut back = dom.coset_table(&g_inv);
        back.par_iter_mut().for_each(|t| *t *= z_inv);
        dom.ifft_in_to_out(&mut ab);
        check(unsafe { sys::csh_vec_mul_table(curve_id::<P>(), limbs_mut(&mut ab), limbs_of(&back), domain_size, 1) })?;
        check(unsafe { sys::csh_bit_reverse(curve_id::<P>(), limbs_mut(&mut ab), log_n, 1) })?; // natural-order coefficients
        Ok(ab)
    }
}

/// Device-resident variant for callers that keep the constraint matrices on the GPU (`csh_matrix_upload` once per circuit):
/// the whole of reduction.rs:77-193 -- rows, six transforms, two local multiplications, three coset shifts -- in ONE call with
/// no PCIe traffic besides the witness shares in and h out. `protocol` 0 = plain / Shamir, 1 = Rep3; `mask_seeds` = the party's
/// two ChaCha12 keys and the number of 32-byte chunks already drawn from each (needs an accessor on `Rep3Rand`, whose
/// `rng1` / `rng2` fields are private upstream: rngs.rs:83-86; without it use the trait path above).
#[allow(clippy::too_many_arguments)]
pub fn witness_map_device<F: PrimeField>(
    curve: i32,
    dom: &HipDomain,
    coset_shift: &F,
    protocol: i32,
    party_id: i32,
    a: sys::CshMatrix,
    b: sys::CshMatrix,
    num_constraints: usize,
    public_inputs: &[F],
    witness_shares: *const u64,
    n_witness: usize,
    mask_seeds: Option<(&[u8; 32], u64, &[u8; 32], u64)>,
    h_out: &mut [F],
) -> eyre::Result<()> {
    let _ = curve;
    let (s1, o1, s2, o2) = match mask_seeds {
        Some((s1, o1, s2, o2)) => (s1.as_ptr(), o1, s2.as_ptr(), o2),
        None => (core::ptr::null(), 0, core::ptr::null(), 0),
    };
    check(unsafe {
        sys::csh_groth16_witness_map(dom.raw, limbs(coset_shift), protocol, party_id, a, b, num_constraints, limbs_of(public_inputs),
                                     public_inputs.len(), witness_shares, n_witness, s1, o1, s2, o2, limbs_mut(h_out))
    })
}
