//! Reference-side binding for libpsdcascade.so (include/psdcascade.h).
//!
//! NOT compiled in this repository's image (no rustc): this is the shim a
//! stabilizer-stream maintainer drops into `src/psd_gpu.rs` and re-exports from
//! `src/lib.rs` in place of `psd::PsdCascade`.  `src/bin/psd.rs` and
//! `src/bin/stream_test.rs` then build unchanged: same type name, same const
//! generic, same method signatures, same panics-on-misuse convention.
//!
//! build.rs:  println!("cargo:rustc-link-lib=dylib=psdcascade");
//!            println!("cargo:rustc-link-search=native={}", env!("PSDCASCADE_LIB_DIR"));

use std::ffi::CStr;
use std::ops::Range;
use std::os::raw::{c_char, c_int};
use std::ptr::NonNull;

pub use crate::psd::{AvgOpts, Break, Detrend, MergeOpts}; // plain data types stay the reference's own

#[repr(C)]
struct PsdcHandle {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Default, Clone, Copy)]
struct PsdcBreak {
    start: u64,
    include: u32,
    count: u32,
    avg: u32,
    _pad: u32,
    bins_start: u64,
    bins_end: u64,
    fft_size: u64,
    decimation: u64,
    pending: u64,
    processed: u64,
}

extern "C" {
    fn psdc_create(n: u32, window_kind: c_int, n_channels: u32, device: c_int) -> *mut PsdcHandle;
    fn psdc_destroy(h: *mut PsdcHandle);
    fn psdc_clone(h: *mut PsdcHandle) -> *mut PsdcHandle;
    fn psdc_set_detrend(h: *mut PsdcHandle, kind: c_int) -> c_int;
    fn psdc_set_avg(h: *mut PsdcHandle, limit: u32, count: u32) -> c_int;
    fn psdc_process(h: *mut PsdcHandle, channel: u32, x: *const f32, len: usize) -> c_int;
    fn psdc_num_stages(h: *mut PsdcHandle, channel: u32) -> c_int;
    fn psdc_psd(
        h: *mut PsdcHandle, channel: u32, keep_overlap: c_int, min_count: u32, keep_transition_band: c_int,
        psd_out: *mut f32, psd_cap: usize, psd_len: *mut usize,
        breaks: *mut PsdcBreak, breaks_cap: usize, n_breaks: *mut usize,
    ) -> c_int;
    fn psdc_rbw(h: *const PsdcHandle) -> f32;
    fn psdc_last_error(h: *const PsdcHandle) -> *const c_char;
}

/// Online power spectral density estimation on one MI355X (drop-in for `psd::PsdCascade<N>`).
pub struct PsdCascade<const N: usize>(NonNull<PsdcHandle>);

// One cascade is used from one thread at a time and is moved into the receiver
// thread (src/bin/psd.rs:168-176): Send, not Sync.
unsafe impl<const N: usize> Send for PsdCascade<N> {}

impl<const N: usize> PsdCascade<N> {
    fn check(&self, rc: c_int) {
        if rc < 0 {
            let msg = unsafe { CStr::from_ptr(psdc_last_error(self.0.as_ptr())) };
            panic!("psdcascade: {}", msg.to_string_lossy()); // the reference panics on misuse
        }
    }

    /// Resolution bandwidth (relative), src/psd.rs:427-429
    pub fn rbw(&self) -> f32 {
        unsafe { psdc_rbw(self.0.as_ptr()) }
    }

    pub fn set_avg(&mut self, avg: AvgOpts) {
        self.check(unsafe { psdc_set_avg(self.0.as_ptr(), avg.limit, avg.count) });
    }

    pub fn set_detrend(&mut self, d: Detrend) {
        // Detrend::Linear returns PSDC_ERR_UNIMPLEMENTED -> panic, like unimplemented!() (src/psd.rs:110)
        self.check(unsafe { psdc_set_detrend(self.0.as_ptr(), d as c_int) });
    }

    /// Process input items (src/psd.rs:456)
    pub fn process(&mut self, x: &[f32]) {
        self.check(unsafe { psdc_process(self.0.as_ptr(), 0, x.as_ptr(), x.len()) });
    }

    /// Return the PSD and a Vec of segment break information (src/psd.rs:479)
    pub fn psd(&self, opts: &MergeOpts) -> (Vec<f32>, Vec<Break>) {
        let h = self.0.as_ptr();
        let ns = unsafe { psdc_num_stages(h, 0) };
        self.check(ns);
        let ns = ns as usize;
        let mut p = vec![0f32; ns * (N / 2 + 1)];
        let mut b = vec![PsdcBreak::default(); ns];
        let (mut plen, mut nb) = (0usize, 0usize);
        self.check(unsafe {
            psdc_psd(h, 0, opts.keep_overlap as c_int, opts.min_count, opts.keep_transition_band as c_int,
                     p.as_mut_ptr(), p.len(), &mut plen, b.as_mut_ptr(), b.len(), &mut nb)
        });
        p.truncate(plen);
        let breaks = b[..nb]
            .iter()
            .map(|b| Break {
                start: b.start as usize,
                include: b.include != 0,
                count: b.count,
                avg: b.avg,
                bins: Range { start: b.bins_start as usize, end: b.bins_end as usize },
                fft_size: b.fft_size as usize,
                decimation: b.decimation as usize,
                pending: b.pending as usize,
                processed: b.processed as usize,
            })
            .collect();
        (p, breaks)
    }
}

impl<const N: usize> Default for PsdCascade<N> {
    /// Hann window, Detrend::None, AvgOpts::default() (src/psd.rs:408-423)
    fn default() -> Self {
        const HANN: c_int = 1;
        let h = unsafe { psdc_create(N as u32, HANN, 1, 0) };
        match NonNull::new(h) {
            Some(h) => Self(h),
            None => {
                let msg = unsafe { CStr::from_ptr(psdc_last_error(std::ptr::null())) };
                panic!("psdcascade: {}", msg.to_string_lossy())
            }
        }
    }
}

impl<const N: usize> Clone for PsdCascade<N> {
    fn clone(&self) -> Self {
        let h = unsafe { psdc_clone(self.0.as_ptr()) };
        Self(NonNull::new(h).expect("psdc_clone failed"))
    }
}

impl<const N: usize> Drop for PsdCascade<N> {
    fn drop(&mut self) {
        unsafe { psdc_destroy(self.0.as_ptr()) }
    }
}
