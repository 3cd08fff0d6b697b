//! Status codes of the C ABI -> eyre (fallible seams) or panic (the reference's infallible trait methods: mpc.rs:84-105 return
//! values, not `Result`; arkworks itself aborts on allocation failure, so a panic is the matching behaviour).
use std::ffi::CStr;

pub fn last_error() -> String {
    unsafe { CStr::from_ptr(cosnarks_hip_sys::csh_last_error()) }.to_string_lossy().into_owned()
}

pub fn check(rc: i32) -> eyre::Result<()> {
    match rc {
        0 => Ok(()),
        cosnarks_hip_sys::CSH_ERR_DOMAIN => eyre::bail!("Polynomial Degree too large"), // the message of reduction.rs:87-94
        _ => eyre::bail!("cosnarks_hip error {rc}: {}", last_error()),
    }
}

#[track_caller]
pub fn hip_ok(rc: i32) {
    if rc != 0 {
        panic!("cosnarks_hip error {rc}: {}", last_error());
    }
}
