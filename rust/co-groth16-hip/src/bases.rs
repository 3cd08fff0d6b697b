//! Proving-key queries live on the device: uploaded the first time a slice is seen ON THE CALLING THREAD'S GPU, reused by every
//! later proof on that GPU (`csh_bases_upload` once per key and device; groth16.rs:219-290 passes sub-slices of the same five
//! queries on every call).
//!
//! Ownership: the cache is keyed by (device, host address range) and every entry carries checkpoint hashes of up to 65 evenly
//! spaced points it was uploaded from. A lookup re-hashes ONLY the checkpoints that lie inside the slice the caller passed (memory
//! the caller holds -- never the rest of the entry's old range, which may have been freed), so a proving key that is dropped and
//! another one allocated at the same address is detected and re-uploaded instead of silently proving against stale bases. [`KeyGuard`] ties the entries of one key to a scope: uploads made while a
//! guard for the key's address range is alive are evicted when it drops.
use crate::error::hip_ok;
use crate::layout::{curve_id, group_id};
use ark_ec::pairing::Pairing;
use ark_ec::short_weierstrass::{Affine, SWCurveConfig};
use cosnarks_hip_sys as sys;
use parking_lot::Mutex;
use std::sync::Arc;

pub struct DeviceBases {
    pub handle: sys::CshBases,
    pub device: i32,  // the GPU the handle lives on (`csh_msm*` checks it against the calling thread's device)
    host_base: usize, // address of the first uploaded point: sub-slices of the same query map to an offset
    len: usize,
    stride: usize,
    /// Padding points in front of the query on the device (point i of the host slice is point `lead + i` of the handle). `KeyGuard`
    /// uploads the l query behind `a_query.len() - l_query.len()` of them: the handle then has the length of the a / b queries and takes
    /// `aux_assignment` at their offset, which is what lets table handles share ONE digit sort inside the library when the reference's
    /// four closures call `csh_msm` over the same slice (cosnarks_hip.h, `csh_msm_multi_dev`; `msm_share_uploads` = 2).
    lead: usize,
    checkpoints: Vec<(usize, u64)>, // (point index, FNV-1a of that point as uploaded), ascending
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
        self.offset_of_raw(points.as_ptr() as usize, points.len())
    }
    fn offset_of_raw(&self, a: usize, len: usize) -> Option<usize> {
        let end = self.host_base + self.len * self.stride;
        (a >= self.host_base && a + len * self.stride <= end && (a - self.host_base) % self.stride == 0)
            .then(|| (a - self.host_base) / self.stride)
    }
    fn overlaps(&self, lo: usize, hi: usize) -> bool {
        self.host_base < hi && lo < self.host_base + self.len * self.stride
    }
}

fn point_hash(base: usize, idx: usize, stride: usize) -> u64 {
    // SAFETY: the caller guarantees that point `idx` of the slice starting at `base` is inside a slice it currently holds
    let bytes = unsafe { core::slice::from_raw_parts((base + idx * stride) as *const u8, stride) };
    bytes.iter().fold(0xcbf29ce484222325u64, |h, &b| (h ^ b as u64).wrapping_mul(0x100000001b3))
}

/// Checkpoints of a freshly uploaded slice: every (len / 64)-th point and the last one.
fn checkpoints_of(base: usize, len: usize, stride: usize) -> Vec<(usize, u64)> {
    if len == 0 {
        return Vec::new();
    }
    let step = (len / 64).max(1);
    let mut idx: Vec<usize> = (0..len).step_by(step).collect();
    if *idx.last().unwrap() != len - 1 {
        idx.push(len - 1);
    }
    idx.into_iter().map(|i| (i, point_hash(base, i, stride))).collect()
}

enum Check {
    Same,      // every checkpoint inside the caller's slice still hashes to what was uploaded
    Different, // at least one differs: the entry is stale
    Unknown,   // no checkpoint falls inside the caller's slice (a sub-slice shorter than the checkpoint spacing)
}

impl DeviceBases {
    /// Compare the entry with `points` (a sub-slice of its address range starting at point `off`), reading only `points`.
    fn check(&self, base: usize, len: usize, off: usize) -> Check {
        let mut seen = false;
        for &(i, h) in self.checkpoints.iter().filter(|(i, _)| *i >= off && *i < off + len) {
            seen = true;
            if point_hash(base, i - off, self.stride) != h {
                return Check::Different;
            }
        }
        if seen || (len == 0 && self.len == 0) { Check::Same } else { Check::Unknown }
    }
}

static CACHE: Mutex<Vec<Arc<DeviceBases>>> = Mutex::new(Vec::new());

fn current_device() -> i32 {
    let mut dev = 0i32;
    hip_ok(unsafe { sys::csh_current_device(&mut dev) });
    dev
}

/// The device copy of `points` on the calling thread's GPU (a whole query or a sub-slice of one already uploaded there) and the
/// offset of `points` in it. A query holding flagged points at infinity must have x = y = 0 on them (the zkey convention, which
/// the reference's parser produces); `normalize_infinity` below does that for keys built elsewhere.
pub fn get_or_upload<P: Pairing, C: SWCurveConfig>(points: &[Affine<C>]) -> (Arc<DeviceBases>, usize) {
    get_or_upload_sized::<P, C>(points, points.len())
}

/// The same with the size of the LARGEST query of the proving key as the table-policy hint (all queries of a key must get the
/// same (c, rows) for `csh_msm_multi_dev` to share one digit pass; `KeyGuard::new` passes it).
pub fn get_or_upload_sized<P: Pairing, C: SWCurveConfig>(points: &[Affine<C>], key_points: usize) -> (Arc<DeviceBases>, usize) {
    // SAFETY: `points` is a live slice of `size_of::<Affine<C>>()`-byte points for the duration of the call
    unsafe { get_or_upload_raw(curve_id::<P>(), group_id::<C>(), points.as_ptr().cast(), points.len(), core::mem::size_of::<Affine<C>>(), key_points) }
}

/// The untyped core (co-noir's curves are not pairings: BN254 G1 and Grumpkin come in as `P: HonkCurve`): `len` points of `stride`
/// bytes each at `ptr`, x then y as the first two coordinates (the flag byte behind them is ignored), on curve / group `curve` / `group`.
/// # Safety
/// `[ptr, ptr + len * stride)` must be a live slice the caller holds for the duration of the call.
pub unsafe fn get_or_upload_raw(curve: i32, group: i32, ptr: *const u8, len: usize, stride: usize, key_points: usize) -> (Arc<DeviceBases>, usize) {
    unsafe { get_or_upload_raw_lead(curve, group, ptr, len, stride, key_points, 0) }
}

/// The same with `lead` padding points in front of the device copy (copies of the slice's first points; never read by an MSM, which
/// starts at the returned offset `lead + (offset inside the query)`). A cache hit keeps the lead the entry was uploaded with.
/// # Safety
/// As `get_or_upload_raw`.
pub unsafe fn get_or_upload_raw_lead(curve: i32, group: i32, ptr: *const u8, len: usize, stride: usize, key_points: usize, lead: usize) -> (Arc<DeviceBases>, usize) {
    let dev = current_device();
    let base_addr = ptr as usize;
    let mut cache = CACHE.lock();
    let mut stale = None;
    let mut cacheable = true;
    for (i, b) in cache.iter().enumerate() {
        if b.device != dev {
            continue; // handles are per device: a thread bound to GPU 1 never gets GPU 0's copy
        }
        if let Some(off) = b.offset_of_raw(base_addr, len) {
            match if b.stride == stride { b.check(base_addr, len, off) } else { Check::Different } {
                Check::Same => return (b.clone(), off + b.lead),
                Check::Different => stale = Some(i), // same addresses, other contents: the key this entry came from is gone
                Check::Unknown => cacheable = false,  // too short to verify against the entry: upload it on its own, uncached
            }
            break;
        }
    }
    if let Some(i) = stale {
        cache.swap_remove(i);
    }
    // an entry of another stride or one that only partly overlaps the caller's range is stale by construction: evict it unread
    let (lo, hi) = (base_addr, base_addr + len * stride);
    if cacheable {
        cache.retain(|b| !(b.device == dev && b.overlaps(lo, hi)));
    }
    let mut handle: sys::CshBases = core::ptr::null_mut();
    let lead = if len == 0 { 0 } else { lead };
    if lead == 0 {
        hip_ok(unsafe { sys::csh_bases_upload(curve, group, ptr.cast(), len, stride, &mut handle) });
    } else {
        // SAFETY: the caller holds `[ptr, ptr + len * stride)`
        let src = unsafe { core::slice::from_raw_parts(ptr, len * stride) };
        let mut padded = Vec::with_capacity((lead + len) * stride);
        for i in 0..lead {
            padded.extend_from_slice(&src[(i % len) * stride..(i % len + 1) * stride]);
        }
        padded.extend_from_slice(src);
        hip_ok(unsafe { sys::csh_bases_upload(curve, group, padded.as_ptr().cast(), lead + len, stride, &mut handle) });
    }
    // Fixed-base tables, the library's own policy (csh_bases_table_policy; the C++ mirror's ProvingKey::build_tables asks the same
    // function). Tables are an optimisation: when they do not fit the device the MSM runs on the plain points.
    let (mut c, mut rows) = (0i32, 0i32);
    hip_ok(unsafe { sys::csh_bases_table_policy(key_points, &mut c, &mut rows) });
    if rows >= 2 && cacheable {
        let rc = unsafe { sys::csh_bases_precompute_grouped(handle, c, rows) };
        if rc == sys::CSH_ERR_OOM {
            hip_ok(unsafe { sys::csh_bases_drop_tables(handle) });
        } else {
            hip_ok(rc);
        }
    }
    let b = Arc::new(DeviceBases { handle, device: dev, host_base: base_addr, len, stride, lead, checkpoints: checkpoints_of(base_addr, len, stride) });
    if cacheable {
        cache.push(b.clone());
    }
    (b, lead)
}

/// Scope of one proving key on one GPU: uploads the five queries up front with a common table policy and evicts them on drop.
/// `ProvingKey<P>` (ark-groth16) fields: `a_query`, `b_g1_query`, `b_g2_query`, `h_query`, `l_query` (groth16.rs:219-290).
pub struct KeyGuard {
    device: i32,
    ranges: Vec<(usize, usize)>,
}
impl KeyGuard {
    pub fn new<P, C1, C2>(pk: &ark_groth16::ProvingKey<P>) -> Self
    where
        P: Pairing<G1Affine = Affine<C1>, G2Affine = Affine<C2>>,
        C1: SWCurveConfig,
        C2: SWCurveConfig,
    {
        let big = [pk.a_query.len(), pk.b_g1_query.len(), pk.l_query.len(), pk.h_query.len(), pk.b_g2_query.len()].into_iter().max().unwrap_or(0);
        let mut ranges = Vec::new();
        // the l query goes up padded to the length of the a / b queries (DeviceBases::lead) when the key has the usual shape
        let same_len = pk.a_query.len() == pk.b_g1_query.len() && pk.a_query.len() == pk.b_g2_query.len() && pk.a_query.len() >= pk.l_query.len();
        let l_lead = if same_len { pk.a_query.len() - pk.l_query.len() } else { 0 };
        for (q, lead) in [(&pk.a_query, 0), (&pk.b_g1_query, 0), (&pk.l_query, l_lead), (&pk.h_query, 0)] {
            // SAFETY: `q` is a live slice of the key the caller holds
            unsafe { get_or_upload_raw_lead(curve_id::<P>(), group_id::<C1>(), q.as_ptr().cast(), q.len(), core::mem::size_of::<Affine<C1>>(), big, lead) };
            ranges.push((q.as_ptr() as usize, q.as_ptr() as usize + q.len() * core::mem::size_of::<Affine<C1>>()));
        }
        get_or_upload_sized::<P, C2>(&pk.b_g2_query, big);
        ranges.push((pk.b_g2_query.as_ptr() as usize, pk.b_g2_query.as_ptr() as usize + pk.b_g2_query.len() * core::mem::size_of::<Affine<C2>>()));
        Self { device: current_device(), ranges }
    }
}
impl Drop for KeyGuard {
    fn drop(&mut self) {
        let mut cache = CACHE.lock();
        cache.retain(|b| !(b.device == self.device && self.ranges.iter().any(|&(lo, hi)| b.overlaps(lo, hi))));
    }
}

/// Drop every cached upload on every device.
pub fn clear() {
    CACHE.lock().clear();
}

/// arkworks marks infinity with the flag and leaves x, y unspecified; the library reads x = y = 0 as infinity and ignores the flag.
pub fn normalize_infinity<C: SWCurveConfig>(points: &mut [Affine<C>]) {
    use ark_ff::Zero;
    for p in points.iter_mut().filter(|p| p.infinity) {
        p.x = C::BaseField::zero();
        p.y = C::BaseField::zero();
    }
}
