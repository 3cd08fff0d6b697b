//! Proving-key queries live on the device: uploaded the first time a slice is seen, reused by every later proof
//! (`csh_bases_upload` once per key; groth16.rs:219-290 passes sub-slices of the same five queries on every call).
use crate::error::hip_ok;
use crate::layout::{curve_id, group_id};
use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, SWCurveConfig};
use cosnarks_hip_sys as sys;
use parking_lot::Mutex;
use std::collections::HashMap;
use std::sync::Arc;

pub struct DeviceBases {
    pub handle: sys::CshBases,
    host_base: usize, // address of the first uploaded point: sub-slices of the same query map to an offset
    len: usize,
    stride: usize,
}
unsafe impl Send for DeviceBases {}
unsafe impl Sync for DeviceBases {}
impl Drop for DeviceBases {
    fn drop(&mut self) {
        unsafe { sys::csh_bases_free(self.handle) };
    }
}
impl DeviceBases {
    /// Offset (in points) of `points` inside the uploaded query, if it is a sub-slice of it.
    pub fn offset_of<T>(&self, points: &[T]) -> Option<usize> {
        let a = points.as_ptr() as usize;
        let end = self.host_base + self.len * self.stride;
        (a >= self.host_base && a + points.len() * self.stride <= end && (a - self.host_base) % self.stride == 0)
            .then(|| (a - self.host_base) / self.stride)
    }
}

type Key = (i32 /* device */, usize /* address */, usize /* len */);
static CACHE: Mutex<Option<HashMap<Key, Arc<DeviceBases>>>> = Mutex::new(None);

/// The device copy of `points` (a whole query or a sub-slice of one already uploaded) and the offset of `points` in it.
/// A query holding flagged points at infinity must have x = y = 0 on them (the zkey convention, which the reference's parser
/// produces); `normalize_infinity` below does that for keys built elsewhere.
pub fn get_or_upload<P: Pairing, C: SWCurveConfig>(points: &[Affine<C>]) -> (Arc<DeviceBases>, usize) {
    let mut dev = 0i32;
    hip_ok(unsafe { sys::csh_current_device(&mut dev) });
    let mut guard = CACHE.lock();
    let map = guard.get_or_insert_with(HashMap::new);
    for b in map.values() {
        if let Some(off) = b.offset_of(points) {
            return (b.clone(), off);
        }
    }
    let mut handle: sys::CshBases = core::ptr::null_mut();
    let stride = core::mem::size_of::<Affine<C>>();
    hip_ok(unsafe {
        sys::csh_bases_upload(curve_id::<P>(), group_id::<C>(), points.as_ptr().cast(), points.len(), stride, &mut handle)
    });
    // Proving-key queries are reused across proofs: four-row fixed-base tables (4x the key memory on the device, built once)
    // let windows w, w + W', w + 2W', w + 3W' share a bucket set. Same policy as ProvingKey::build_tables of the C++ mirror:
    // only for 2^14..2^21 points. The width follows the slice length (the cache sees slices, not keys): queries of one key whose
    // lengths straddle a power of two get different widths and csh_msm_multi_dev then runs that call on the plain points.
    if (1usize << 14..=1usize << 21).contains(&points.len()) {
        let mut c = 16i32;
        while c > 10 && (1usize << (c + 1)) > points.len() {
            c -= 1;
        }
        hip_ok(unsafe { sys::csh_bases_precompute_grouped(handle, c, 4) });
    }
    let b = Arc::new(DeviceBases { handle, host_base: points.as_ptr() as usize, len: points.len(), stride });
    map.insert((dev, points.as_ptr() as usize, points.len()), b.clone());
    (b, 0)
}

/// Drop every cached upload (call when a proving key is dropped: the cache is keyed by host addresses).
pub fn clear() {
    *CACHE.lock() = None;
}

/// arkworks marks infinity with the flag and leaves x, y unspecified; the library reads x = y = 0 as infinity and ignores the flag.
pub fn normalize_infinity<C: SWCurveConfig>(points: &mut [Affine<C>]) {
    use ark_ff::Zero;
    for p in points.iter_mut().filter(|p| p.infinity) {
        p.x = C::BaseField::zero();
        p.y = C::BaseField::zero();
    }
}
