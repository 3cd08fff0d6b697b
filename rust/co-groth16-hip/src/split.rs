//! One MSM split over the GPUs of a node (BASELINE config 5): contiguous point ranges, one RCCL all-gather of the window
//! partials, host fold -- `csh_comm_*` + `csh_msm_split_rank_dev` / `csh_msm_split` behind the C ABI (no torch, RCCL is
//! dlopen'ed by the library). Same seam as `msm_public_points_hs`: groth16.rs:227-294.
use crate::error::check;
use cosnarks_hip_sys as sys;

/// One rank (thread or process bound to one GPU with `bind_device`) of a split-MSM communicator.
pub struct SplitComm(sys::CshComm);
unsafe impl Send for SplitComm {}
impl Drop for SplitComm {
    fn drop(&mut self) {
        unsafe { sys::csh_comm_destroy(self.0) };
    }
}
impl SplitComm {
    /// Rank 0 draws the id and ships the 128 bytes to the other ranks over whatever channel the host has (mpc-net, a pipe).
    pub fn unique_id() -> eyre::Result<[u8; sys::CSH_COMM_ID_BYTES]> {
        let mut id = [0u8; sys::CSH_COMM_ID_BYTES];
        check(unsafe { sys::csh_comm_unique_id(id.as_mut_ptr()) })?;
        Ok(id)
    }
    pub fn init_rank(id: &[u8; sys::CSH_COMM_ID_BYTES], nranks: i32, rank: i32) -> eyre::Result<Self> {
        let mut c: sys::CshComm = core::ptr::null_mut();
        check(unsafe { sys::csh_comm_init_rank(id.as_ptr(), nranks, rank, &mut c) })?;
        Ok(Self(c))
    }
    /// This rank's range of the MSM; collective; every rank receives the full result (`out` = arkworks `Projective<C>`).
    pub fn msm<Out>(&self, bases: sys::CshBases, offset: usize, n: usize, scalars_dev: *const u64, out: &mut Out) -> eyre::Result<()> {
        check(unsafe { sys::csh_msm_split_rank_dev(self.0, bases, offset, n, scalars_dev, 1, (out as *mut Out).cast(), core::ptr::null_mut()) })
    }
}

/// One thread driving every GPU: `parts[i]` = (bases handle on its device, offset, count, device scalars).
pub fn msm_split_single_thread<Out>(parts: &[(sys::CshBases, usize, usize, *const u64)], mode: i32, out: &mut Out) -> eyre::Result<()> {
    let bases: Vec<_> = parts.iter().map(|p| p.0).collect();
    let offs: Vec<_> = parts.iter().map(|p| p.1).collect();
    let cnts: Vec<_> = parts.iter().map(|p| p.2).collect();
    let scal: Vec<_> = parts.iter().map(|p| p.3).collect();
    check(unsafe {
        sys::csh_msm_split(bases.as_ptr(), offs.as_ptr(), cnts.as_ptr(), scal.as_ptr(), parts.len(), 1, mode, core::ptr::null(), (out as *mut Out).cast())
    })
}
