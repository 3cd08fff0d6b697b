//! Constraint matrices live on the device: flattened to CSR and uploaded (`csh_matrix_upload`) the first time a
//! `ConstraintMatrices` is seen on the calling thread's GPU, reused by every later proof of the same circuit -- the witness maps
//! (reduction.rs:99-130, 260-298) evaluate the same rows on every call.
//!
//! Keyed like `bases.rs`: (device, address of the row vector, number of rows, non-zeros) plus checkpoint hashes of rows the caller
//! holds, so a dropped circuit whose allocation is reused by another one is detected and re-uploaded, never silently reused.
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

/// The three sides of one circuit on one GPU (`c` only when a LibSnark map asked for it).
pub struct DeviceMatrices {
    pub a: DeviceMatrix,
    pub b: DeviceMatrix,
    pub c: Option<DeviceMatrix>,
    device: i32,
    key: (usize, usize, usize, usize), // (&matrices.a[0], rows of a, &matrices.b[0], rows of b)
    fingerprint: u64,
}

/// FNV-1a over (length, first entry, last entry) of up to 64 evenly spaced rows of both sides: memory the caller holds.
fn fingerprint<F>(a: &Matrix<F>, b: &Matrix<F>) -> u64 {
    let mut h = 0xcbf29ce484222325u64;
    let mut eat = |bytes: &[u8]| {
        for &x in bytes {
            h = (h ^ x as u64).wrapping_mul(0x100000001b3);
        }
    };
    for m in [a, b] {
        let step = (m.len() / 64).max(1);
        for row in m.iter().step_by(step).chain(m.last()) {
            eat(&row.len().to_le_bytes());
            for e in row.first().into_iter().chain(row.last()) {
                // SAFETY: `e` is a live (F, usize) of the caller's matrix; read as plain bytes (F is `Copy` limbs, no padding reads matter to a hash)
                eat(unsafe { core::slice::from_raw_parts((&e.0 as *const F).cast::<u8>(), core::mem::size_of::<F>()) });
                eat(&e.1.to_le_bytes());
            }
        }
    }
    h
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

/// Device copy of `matrices` on the calling thread's GPU; `with_c` also uploads the C side (LibSnarkReduction, reduction.rs:292-298).
pub fn get_or_upload<P: Pairing>(matrices: &ConstraintMatrices<P::ScalarField>, with_c: bool) -> eyre::Result<Arc<DeviceMatrices>> {
    let dev = current_device();
    let key = (matrices.a.as_ptr() as usize, matrices.a.len(), matrices.b.as_ptr() as usize, matrices.b.len());
    let fp = fingerprint(&matrices.a, &matrices.b);
    let mut cache = CACHE.lock();
    if let Some(i) = cache.iter().position(|e| e.device == dev && e.key == key) {
        if cache[i].fingerprint == fp && (!with_c || cache[i].c.is_some()) {
            return Ok(cache[i].clone());
        }
        cache.remove(i); // same addresses, other contents (or the C side is missing): re-upload
    }
    let e = Arc::new(DeviceMatrices {
        a: upload_side::<P>(&matrices.a)?,
        b: upload_side::<P>(&matrices.b)?,
        c: if with_c { Some(upload_side::<P>(&matrices.c)?) } else { None },
        device: dev,
        key,
        fingerprint: fp,
    });
    if cache.len() >= CACHE_CIRCUITS {
        cache.remove(0);
    }
    cache.push(e.clone());
    Ok(e)
}

/// Drop every cached circuit on every device.
pub fn clear() {
    CACHE.lock().clear();
}
