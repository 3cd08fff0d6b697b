//! Constraint matrices live on the device: flattened to CSR and uploaded (`csh_matrix_upload`) the first time a
//! `ConstraintMatrices` is seen on the calling thread's GPU, reused by every later proof of the same circuit -- the witness maps
//! (reduction.rs:99-130, 260-298) evaluate the same rows on every call.
//!
//! An entry is reused only when it is KNOWN to describe the caller's matrices (a sampled fingerprint of A and B alone could
//! prove against stale device matrices after an in-place edit, or after another circuit was allocated at the same addresses):
//! * key = (device, address and row count of A, B **and C**, `num_constraints`, `num_instance_variables`, `num_witness_variables`);
//! * by default every lookup re-hashes the FULL contents of every side that is resident (all rows, all entries, positions included;
//!   rayon over the rows, one 64-bit multiply-xorshift per word: ~80 MB at 2^20 constraints, a few ms on the cores rayon has) and
//!   compares with the hash taken from the same memory at upload;
//! * a caller that proves the same circuit repeatedly holds a [`MatricesGuard`]: it keeps a shared borrow of the `ConstraintMatrices`
//!   for its lifetime, so the borrow checker itself rules out in-place mutation, and lookups through the guard skip the re-hash;
//!   dropping the guard evicts the entry. [`invalidate`] evicts explicitly.
use crate::error::check;
use crate::layout::curve_id;
use ark_ec::pairing::Pairing;
use ark_relations::utils::matrix::Matrix;
use cosnarks_hip_sys as sys;
use parking_lot::Mutex;
use std::sync::Arc;
use taceo_groth16::ConstraintMatrices;

pub struct DeviceMatrix {
    pub handle: sys::CshMatrix,
}
unsafe impl Send for DeviceMatrix {}
unsafe impl Sync for DeviceMatrix {}
impl Drop for DeviceMatrix {
    fn drop(&mut self) {
        unsafe { sys::csh_matrix_free(self.handle) };
    }
}

/// (address, rows) of one side; (0, 0) for an empty matrix
type SideKey = (usize, usize);
#[derive(Clone, Copy, PartialEq, Eq, Debug)]
struct Key {
    a: SideKey,
    b: SideKey,
    c: SideKey,
    num_constraints: usize,
    num_instance_variables: usize,
    num_witness_variables: usize,
}
fn key_of<F>(m: &ConstraintMatrices<F>) -> Key {
    let side = |x: &Matrix<F>| (x.as_ptr() as usize, x.len());
    Key {
        a: side(&m.a),
        b: side(&m.b),
        c: side(&m.c),
        num_constraints: m.num_constraints,
        num_instance_variables: m.num_instance_variables,
        num_witness_variables: m.num_witness_variables,
    }
}

/// The three sides of one circuit on one GPU (`c` only when a LibSnark map asked for it).
pub struct DeviceMatrices {
    pub a: DeviceMatrix,
    pub b: DeviceMatrix,
    pub c: Option<DeviceMatrix>,
    device: i32,
    key: Key,
    /// full content hash of A, B and (when resident) C, taken from the caller's memory when the sides were uploaded
    content: [u64; 3],
}

#[inline]
fn mix(h: u64, w: u64) -> u64 {
    let x = (h ^ w).wrapping_mul(0x9E3779B97F4A7C15);
    x ^ (x >> 29)
}

/// Hash of every entry of every row (coefficient limbs + column index) and of the row boundaries. Rows are hashed independently
/// (rayon) and folded with their position, so a permutation of rows or a moved entry changes the result.
fn content_hash<F: Sync>(m: &Matrix<F>) -> u64 {
    use rayon::prelude::*;
    let words = core::mem::size_of::<F>() / 8;
    assert_eq!(words * 8, core::mem::size_of::<F>(), "field elements are whole 64-bit limbs (layout.rs)");
    m.par_iter()
        .enumerate()
        .with_min_len(1024)
        .map(|(i, row)| {
            let mut h = mix(0x243F6A8885A308D3, row.len() as u64);
            for (v, idx) in row {
                // SAFETY: `v` is a live F of the caller's matrix, `words` u64 limbs without padding (layout.rs asserts the size)
                let limbs = unsafe { core::slice::from_raw_parts((v as *const F).cast::<u64>(), words) };
                for &w in limbs {
                    h = mix(h, w);
                }
                h = mix(h, *idx as u64);
            }
            mix(h, i as u64).wrapping_mul(2 * i as u64 + 1)
        })
        .reduce(|| m.len() as u64, |x, y| x.wrapping_add(y))
}

fn upload_side<P: Pairing>(m: &Matrix<P::ScalarField>) -> eyre::Result<DeviceMatrix> {
    let nnz: usize = m.iter().map(Vec::len).sum();
    let mut row_ptr = Vec::<u64>::with_capacity(m.len() + 1);
    let mut col = Vec::<u32>::with_capacity(nnz);
    let mut coef = Vec::<P::ScalarField>::with_capacity(nnz);
    row_ptr.push(0);
    for row in m {
        for (v, idx) in row {
            col.push(u32::try_from(*idx).map_err(|_| eyre::eyre!("constraint column index {idx} exceeds 32 bits"))?);
            coef.push(*v);
        }
        row_ptr.push(col.len() as u64);
    }
    let mut handle: sys::CshMatrix = core::ptr::null_mut();
    check(unsafe { sys::csh_matrix_upload(curve_id::<P>(), row_ptr.as_ptr(), col.as_ptr(), coef.as_ptr().cast(), m.len(), nnz, &mut handle) })?;
    Ok(DeviceMatrix { handle })
}

static CACHE: Mutex<Vec<Arc<DeviceMatrices>>> = Mutex::new(Vec::new());
const CACHE_CIRCUITS: usize = 8; // per process; the oldest entry goes first

fn current_device() -> i32 {
    let mut dev = 0i32;
    crate::error::hip_ok(unsafe { sys::csh_current_device(&mut dev) });
    dev
}

fn lookup<P: Pairing>(matrices: &ConstraintMatrices<P::ScalarField>, with_c: bool, trusted: bool) -> eyre::Result<Arc<DeviceMatrices>> {
    let dev = current_device();
    let key = key_of(matrices);
    // hash outside the lock: what the caller holds NOW (skipped for a guarded circuit: the guard's borrow forbids mutation)
    let now = if trusted { None } else { Some([content_hash(&matrices.a), content_hash(&matrices.b), if with_c { content_hash(&matrices.c) } else { 0 }]) };
    let mut cache = CACHE.lock();
    if let Some(i) = cache.iter().position(|e| e.device == dev && e.key == key) {
        let e = &cache[i];
        let have_c = e.c.is_some();
        let same = match now {
            None => true,
            Some(h) => h[0] == e.content[0] && h[1] == e.content[1] && (!with_c || !have_c || h[2] == e.content[2]),
        };
        if same && (!with_c || have_c) {
            return Ok(cache[i].clone());
        }
        cache.remove(i); // same addresses, other contents (or the C side is missing): re-upload
    }
    drop(cache); // uploads take milliseconds: not under the lock
    let content = match now {
        Some(h) => [h[0], h[1], if with_c { h[2] } else { 0 }],
        None => [content_hash(&matrices.a), content_hash(&matrices.b), if with_c { content_hash(&matrices.c) } else { 0 }],
    };
    let e = Arc::new(DeviceMatrices {
        a: upload_side::<P>(&matrices.a)?,
        b: upload_side::<P>(&matrices.b)?,
        c: if with_c { Some(upload_side::<P>(&matrices.c)?) } else { None },
        device: dev,
        key,
        content,
    });
    let mut cache = CACHE.lock();
    cache.retain(|x| !(x.device == dev && x.key == key)); // another thread may have uploaded the same circuit meanwhile
    if cache.len() >= CACHE_CIRCUITS {
        cache.remove(0);
    }
    cache.push(e.clone());
    Ok(e)
}

/// keys of the circuits a live [`MatricesGuard`] vouches for (process-wide: a prover's witness maps run on rayon workers)
static GUARDED: Mutex<Vec<Key>> = Mutex::new(Vec::new());

/// Device copy of `matrices` on the calling thread's GPU; `with_c` also uploads the C side (LibSnarkReduction, reduction.rs:292-298).
/// Verified against the caller's memory on every call unless a [`MatricesGuard`] for these matrices is alive.
pub fn get_or_upload<P: Pairing>(matrices: &ConstraintMatrices<P::ScalarField>, with_c: bool) -> eyre::Result<Arc<DeviceMatrices>> {
    let key = key_of(matrices);
    let trusted = GUARDED.lock().contains(&key);
    lookup::<P>(matrices, with_c, trusted)
}

/// Scope of one circuit's device copy. Holds a shared borrow of the matrices, so they can be neither mutated nor dropped while it
/// lives: witness maps of this circuit skip the per-call content hash. Dropping it evicts the circuit from every device.
pub struct MatricesGuard<'a, F> {
    matrices: &'a ConstraintMatrices<F>,
    key: Key,
}
impl<'a, F> MatricesGuard<'a, F> {
    pub fn new(matrices: &'a ConstraintMatrices<F>) -> Self {
        let key = key_of(matrices);
        CACHE.lock().retain(|e| e.key != key); // whatever is cached under these addresses predates the guard: not vouched for
        GUARDED.lock().push(key);
        Self { matrices, key }
    }
    pub fn matrices(&self) -> &'a ConstraintMatrices<F> {
        self.matrices
    }
}
impl<F> Drop for MatricesGuard<'_, F> {
    fn drop(&mut self) {
        {
            let mut g = GUARDED.lock();
            if let Some(i) = g.iter().position(|k| *k == self.key) {
                g.swap_remove(i);
            }
        }
        let key = self.key;
        CACHE.lock().retain(|e| e.key != key);
    }
}

/// Evict the device copies of `matrices` (every device): the next witness map uploads them again.
pub fn invalidate<F>(matrices: &ConstraintMatrices<F>) {
    let key = key_of(matrices);
    CACHE.lock().retain(|e| e.key != key);
}

/// Drop every cached circuit on every device.
pub fn clear() {
    CACHE.lock().clear();
}
