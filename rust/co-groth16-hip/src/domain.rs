//! `taceo_ark_algebra::fft::Domain` on the device: `csh_domain_create` + the four transforms (reduction.rs:10, 93, 141-174).
use crate::error::{check, hip_ok};
use crate::layout::{limbs, limbs_mut, ncomp};
use cosnarks_hip_sys as sys;

pub struct HipDomain {
    pub(crate) raw: sys::CshDomain,
    pub size: usize,
}
unsafe impl Send for HipDomain {}
unsafe impl Sync for HipDomain {}
impl Drop for HipDomain {
    fn drop(&mut self) {
        unsafe { sys::csh_domain_free(self.raw) };
    }
}
/// Domains are reused across proofs: building one is a hipMalloc + a kernel, and a proof of the same circuit asks for the same
/// (device, curve, size, generator) every time. Entries are shared (`Arc`), freed by `clear_cache`.
static DOMAINS: parking_lot::Mutex<Vec<((i32, i32, u32, [u64; 4]), std::sync::Arc<HipDomain>)>> = parking_lot::Mutex::new(Vec::new());

pub fn clear_cache() {
    DOMAINS.lock().clear();
}

impl HipDomain {
    /// `new` through the process-wide cache, keyed by (device, curve, log size, generator limbs; all-zero = arkworks' default root).
    pub fn cached<F>(curve: i32, log_n: u32, group_gen: Option<&F>) -> eyre::Result<std::sync::Arc<Self>> {
        assert!(core::mem::size_of::<F>() == 32, "scalar fields are four 64-bit limbs");
        let mut dev = 0i32;
        check(unsafe { sys::csh_current_device(&mut dev) })?;
        let gen_limbs: [u64; 4] = group_gen.map_or([0; 4], |g| unsafe { core::ptr::read(limbs(g).cast()) });
        let key = (dev, curve, log_n, gen_limbs);
        let mut cache = DOMAINS.lock();
        if let Some((_, d)) = cache.iter().find(|(k, _)| *k == key) {
            return Ok(d.clone());
        }
        let d = std::sync::Arc::new(Self::new(curve, log_n, group_gen)?);
        cache.push((key, d.clone()));
        Ok(d)
    }
    /// `Domain::with_group_gen` (snarkjs roots, reduction.rs:93) when `group_gen` is given, `Domain::new` otherwise (:249).
    pub fn new<F>(curve: i32, log_n: u32, group_gen: Option<&F>) -> eyre::Result<Self> {
        let mut raw: sys::CshDomain = core::ptr::null_mut();
        let g = group_gen.map_or(core::ptr::null(), |g| limbs(g));
        check(unsafe { sys::csh_domain_create(curve, log_n, g, &mut raw) })?;
        Ok(Self { raw, size: 1usize << log_n })
    }
    /// The C ABI handle (for sibling crates that call `csh_fft` / `csh_ifft` on it).
    pub fn raw(&self) -> sys::CshDomain {
        self.raw
    }
    /// natural-order evaluations -> bit-reversed coefficients (scaled by 1/n); `S` = F, a Shamir share or a Rep3 share
    pub fn ifft_in_to_out<S>(&self, data: &mut [S]) {
        assert_eq!(data.len(), self.size);
        hip_ok(unsafe { sys::csh_ifft_in_to_out(self.raw, limbs_mut(data), ncomp::<S>()) });
    }
    pub fn fft_out_to_in<S>(&self, data: &mut [S]) {
        assert_eq!(data.len(), self.size);
        hip_ok(unsafe { sys::csh_fft_out_to_in(self.raw, limbs_mut(data), ncomp::<S>()) });
    }
    /// `bit_reversed_coset_table(shift, size)` (reduction.rs:45-60)
    pub fn coset_table<F: Clone + Default>(&self, shift: &F) -> Vec<F> {
        let mut t = vec![F::default(); self.size];
        hip_ok(unsafe { sys::csh_coset_table(self.raw, limbs(shift), limbs_mut(&mut t)) });
        t
    }
}
