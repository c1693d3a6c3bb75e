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

use rustfft::Fft;
use std::ffi::CStr;
use std::ops::Range;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr::NonNull;
use std::sync::Arc;

pub use crate::psd::{AvgOpts, Break, Detrend, MergeOpts, Window}; // plain data types stay the reference's own

/// `device` argument meaning "the index in $PSDC_DEVICE, 0 when unset" (include/psdcascade.h): what the constructors
/// with the reference's signatures pass, so that one process per GPU is placed from outside (`PSDC_DEVICE=3 psd ...`).
const PSDC_DEVICE_DEFAULT: c_int = -1;

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
    fn psdc_create_window(n: u32, win: *const f32, power: f32, nenbw: f32, overlap: usize, n_channels: u32,
                          device: c_int) -> *mut PsdcHandle;
    fn psdc_readout_bytes(n: u32, n_channels: u32) -> usize;
    fn psdc_pack_readout(h: *mut PsdcHandle, buf: *mut c_void, cap: usize, len: *mut usize) -> c_int;
    fn psdc_pack_pad(rec: *const c_void, len: usize, out: *mut c_void, cap: usize, n_channels: u32) -> c_int;
    fn psdc_unpack_stitch(
        buf: *const c_void, len: usize, channel: u32, keep_overlap: c_int, min_count: u32, keep_transition_band: c_int,
        psd_out: *mut f32, psd_cap: usize, psd_len: *mut usize,
        breaks: *mut PsdcBreak, breaks_cap: usize, n_breaks: *mut usize,
    ) -> c_int;
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
        (p, b[..nb].iter().map(to_break).collect())
    }

    /// A cascade on a given HIP device, or over a caller-built window: the reference has neither constructor
    /// (`Default` plans Hann on the CPU, src/psd.rs:408-423); `default()` == `with(Window::hann(), PSDC_DEVICE)`.
    pub fn with_device(device: i32) -> Self {
        const HANN: c_int = 1;
        Self(nonnull_or_panic(unsafe { psdc_create(N as u32, HANN, 1, device as c_int) }))
    }
    pub fn with_window(win: &Window<N>, device: i32) -> Self {
        Self(nonnull_or_panic(unsafe {
            psdc_create_window(N as u32, win.win.as_ptr(), win.power, win.nenbw, win.overlap, 1, device as c_int)
        }))
    }

    /// The raw accumulators and counters of this cascade as a flat, fixed-size byte record
    /// (`Self::READOUT_BYTES`, the same for every cascade of this N): what a multi-GPU host gathers -- with RCCL,
    /// MPI, a socket or a memcpy between the handles of one process -- and stitches with [`psd_from_readout`].
    pub fn pack_readout(&self) -> Vec<u8> {
        let mut buf = vec![0u8; unsafe { psdc_readout_bytes(N as u32, 1) }];
        let mut len = 0usize;
        self.check(unsafe { psdc_pack_readout(self.0.as_ptr(), buf.as_mut_ptr() as *mut c_void, buf.len(), &mut len) });
        buf
    }
}

fn nonnull_or_panic(h: *mut PsdcHandle) -> NonNull<PsdcHandle> {
    NonNull::new(h).unwrap_or_else(|| {
        let msg = unsafe { CStr::from_ptr(psdc_last_error(std::ptr::null())) };
        panic!("psdcascade: {}", msg.to_string_lossy())
    })
}

fn to_break(b: &PsdcBreak) -> Break {
    Break {
        start: b.start as usize,
        include: b.include != 0,
        count: b.count,
        avg: b.avg,
        bins: Range { start: b.bins_start as usize, end: b.bins_end as usize },
        fft_size: b.fft_size as usize,
        decimation: b.decimation as usize,
        pending: b.pending as usize,
        processed: b.processed as usize,
    }
}

/// `PsdCascade::psd` (src/psd.rs:479-543) of one gathered record: bit-identical to `psd()` on the cascade that
/// packed it (raw accumulators travel; the 64-bit counts the library normalises by travel with them).
pub fn psd_from_readout<const N: usize>(record: &[u8], opts: &MergeOpts) -> (Vec<f32>, Vec<Break>) {
    let mut p = vec![0f32; 16 * (N / 2 + 1)];
    let mut b = vec![PsdcBreak::default(); 16];
    let (mut plen, mut nb) = (0usize, 0usize);
    let rc = unsafe {
        psdc_unpack_stitch(record.as_ptr() as *const c_void, record.len(), 0, opts.keep_overlap as c_int, opts.min_count,
                           opts.keep_transition_band as c_int, p.as_mut_ptr(), p.len(), &mut plen, b.as_mut_ptr(),
                           b.len(), &mut nb)
    };
    if rc < 0 {
        let msg = unsafe { CStr::from_ptr(psdc_last_error(std::ptr::null())) };
        panic!("psdcascade: {}", msg.to_string_lossy());
    }
    p.truncate(plen);
    (p, b[..nb].iter().map(to_break).collect())
}

/// A gathered block must have the same size on every shard: `record` as a record of `n_channels` channels, the added
/// ones empty (`psdc_pack_pad`; shards whose channel counts differ pad to the largest).
pub fn pad_readout<const N: usize>(record: &[u8], n_channels: u32) -> Vec<u8> {
    let mut out = vec![0u8; unsafe { psdc_readout_bytes(N as u32, n_channels) }];
    let rc = unsafe {
        psdc_pack_pad(record.as_ptr() as *const c_void, record.len(), out.as_mut_ptr() as *mut c_void, out.len(), n_channels)
    };
    if rc < 0 {
        let msg = unsafe { CStr::from_ptr(psdc_last_error(std::ptr::null())) };
        panic!("psdcascade: {}", msg.to_string_lossy());
    }
    out
}

impl<const N: usize> Default for PsdCascade<N> {
    /// Hann window, Detrend::None, AvgOpts::default() (src/psd.rs:408-423), on the device $PSDC_DEVICE names
    fn default() -> Self {
        Self::with_device(PSDC_DEVICE_DEFAULT)
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

// ---------------------------------------------------------------------------------------------
// PsdBank<N>: all traces of a stream in ONE handle -- not a reference type; the three-line change to the receiver loop of
// src/bin/psd.rs:174-182 / src/bin/stream_test.rs:46-55 that gets the batched path the kernels are tuned for (one round of
// launches for all traces instead of one handle, stream and launch sequence per trace):
//
//     let mut dec = PsdBank::<{ 1 << 9 }>::new(4);                    // was: Vec<(&str, PsdCascade<_>)>, one per trace
//     for (i, (_name, trace)) in traces.iter().enumerate() { dec.process(i, trace); }   // was: dec[i].1.process(trace)
//     let (p, b) = dec.psd(i, &merge_opts);                           // was: dec[i].1.psd(&merge_opts)
// ---------------------------------------------------------------------------------------------

extern "C" {
    fn psdc_process_adcdac_frames(h: *mut PsdcHandle, frames: *const u8, frame_size: usize, n_frames: usize, n_ok: *mut usize) -> c_int;
    fn psdc_process_frames(h: *mut PsdcHandle, frames: *const u8, frame_size: usize, n_frames: usize, n_ok: *mut usize) -> c_int;
    fn psdc_process_frames_device(h: *mut PsdcHandle, d_frames: *const u8, frame_size: usize, n_frames: usize, n_ok: *mut usize) -> c_int;
}

pub struct PsdBank<const N: usize> {
    h: NonNull<PsdcHandle>,
    n_traces: usize,
}

unsafe impl<const N: usize> Send for PsdBank<N> {}

impl<const N: usize> PsdBank<N> {
    pub fn new(n_traces: usize) -> Self {
        const HANN: c_int = 1;
        Self { h: nonnull_or_panic(unsafe { psdc_create(N as u32, HANN, n_traces as u32, PSDC_DEVICE_DEFAULT) }), n_traces }
    }
    fn check(&self, rc: c_int) {
        if rc < 0 {
            let msg = unsafe { CStr::from_ptr(psdc_last_error(self.h.as_ptr())) };
            panic!("psdcascade: {}", msg.to_string_lossy());
        }
    }
    pub fn set_avg(&mut self, avg: AvgOpts) {
        self.check(unsafe { psdc_set_avg(self.h.as_ptr(), avg.limit, avg.count) });
    }
    pub fn set_detrend(&mut self, d: Detrend) {
        self.check(unsafe { psdc_set_detrend(self.h.as_ptr(), d as c_int) });
    }
    /// `PsdCascade::process` of trace `i`
    pub fn process(&mut self, i: usize, x: &[f32]) {
        assert!(i < self.n_traces);
        self.check(unsafe { psdc_process(self.h.as_ptr(), i as u32, x.as_ptr(), x.len()) });
    }
    /// whole frames as `Source::get` reads them for `Data::File` / `Data::Udp` (src/source.rs:135-142,158-165): header checks,
    /// `Loss` counting and `AdcDac::traces()` of src/de happen behind the call, the four traces land in traces 0..3.
    /// Err(code) carries the `de::Error` of the first bad frame; the frames before it were ingested.
    pub fn process_adcdac_frames(&mut self, frames: &[u8], frame_size: usize) -> Result<usize, i32> {
        let mut ok = 0usize;
        let rc = unsafe { psdc_process_adcdac_frames(self.h.as_ptr(), frames.as_ptr(), frame_size, frames.len() / frame_size, &mut ok) };
        if rc < 0 { Err(rc) } else { Ok(ok) }
    }
    /// the same for any of the four `Format`s (src/de/mod.rs:12-17: `AdcDac`, `Fls`, `ThermostatEem`, `Mpll`), each frame's header
    /// naming its own: trace `i` of `Payload::traces()` lands in trace `i` of the bank, as `dec[i]` in src/bin/psd.rs:174-182.
    pub fn process_frames(&mut self, frames: &[u8], frame_size: usize) -> Result<usize, i32> {
        let mut ok = 0usize;
        let rc = unsafe { psdc_process_frames(self.h.as_ptr(), frames.as_ptr(), frame_size, frames.len() / frame_size, &mut ok) };
        if rc < 0 { Err(rc) } else { Ok(ok) }
    }
    /// `process_frames` for `n_frames` frames that already sit in device memory at `d_frames` (valid and unmodified until `sync()`
    /// or a read-out; see include/psdcascade.h).
    ///
    /// # Safety
    /// `d_frames` must point to `n_frames * frame_size` readable bytes of device memory on the bank's device.
    pub unsafe fn process_frames_device(&mut self, d_frames: *const u8, frame_size: usize, n_frames: usize) -> Result<usize, i32> {
        let mut ok = 0usize;
        let rc = psdc_process_frames_device(self.h.as_ptr(), d_frames, frame_size, n_frames, &mut ok);
        if rc < 0 { Err(rc) } else { Ok(ok) }
    }
    /// `PsdCascade::psd` of trace `i`
    pub fn psd(&self, i: usize, opts: &MergeOpts) -> (Vec<f32>, Vec<Break>) {
        let h = self.h.as_ptr();
        let ns = unsafe { psdc_num_stages(h, i as u32) };
        self.check(ns);
        let ns = ns as usize;
        let mut p = vec![0f32; ns * (N / 2 + 1)];
        let mut b = vec![PsdcBreak::default(); ns];
        let (mut plen, mut nb) = (0usize, 0usize);
        self.check(unsafe {
            psdc_psd(h, i as u32, opts.keep_overlap as c_int, opts.min_count, opts.keep_transition_band as c_int,
                     p.as_mut_ptr(), p.len(), &mut plen, b.as_mut_ptr(), b.len(), &mut nb)
        });
        p.truncate(plen);
        (p, b[..nb].iter().map(to_break).collect())
    }
}

impl<const N: usize> Drop for PsdBank<N> {
    fn drop(&mut self) {
        unsafe { psdc_destroy(self.h.as_ptr()) }
    }
}

// ---------------------------------------------------------------------------------------------
// Psd<N>: one stage with the reference's `PsdStage` trait (src/psd.rs:122-288), e.g. for the crate's
// own test (src/psd.rs:615-632).  `pub use psd_gpu::Psd;` next to `PsdCascade` in src/lib.rs.
// ---------------------------------------------------------------------------------------------

pub use crate::psd::PsdStage; // the trait stays the reference's own (src/psd.rs:163-193)

#[repr(C)]
struct PsdcStage {
    _private: [u8; 0],
}

extern "C" {
    fn psdc_stage_create_window(n: u32, win: *const f32, power: f32, nenbw: f32, overlap: usize, device: c_int) -> *mut PsdcStage;
    fn psdc_stage_destroy(s: *mut PsdcStage);
    fn psdc_stage_clone(s: *mut PsdcStage) -> *mut PsdcStage;
    fn psdc_stage_set_avg(s: *mut PsdcStage, avg: u32) -> c_int;
    fn psdc_stage_set_detrend(s: *mut PsdcStage, kind: c_int) -> c_int;
    fn psdc_stage_process(s: *mut PsdcStage, x: *const f32, len: usize, y: *mut f32, cap: usize, n_out: *mut usize) -> c_int;
    fn psdc_stage_get_spectrum(s: *mut PsdcStage, out: *mut f32) -> c_int;
    fn psdc_stage_get_count(s: *mut PsdcStage, count: *mut u32) -> c_int;
    fn psdc_stage_get_gain(s: *mut PsdcStage, gain: *mut f32) -> c_int;
    fn psdc_stage_get_buf(s: *mut PsdcStage, out: *mut f32, cap: usize, len: *mut usize) -> c_int;
    fn psdc_stage_last_error(s: *const PsdcStage) -> *const c_char;
}

/// Power spectral density accumulator and decimator on one MI355X (drop-in for `psd::Psd<N>`).
///
/// `spectrum()` and `buf()` return slices in the reference; the accumulators live in HBM here, so the
/// shim keeps host copies that are refreshed by `process()` (the only call that changes them).
pub struct Psd<const N: usize> {
    s: NonNull<PsdcStage>,
    spectrum: Vec<f32>, // N/2 + 1
    buf: Vec<f32>,      // pending input items
}

unsafe impl<const N: usize> Send for Psd<N> {}

impl<const N: usize> Psd<N> {
    fn check(&self, rc: c_int) {
        if rc < 0 {
            let msg = unsafe { CStr::from_ptr(psdc_stage_last_error(self.s.as_ptr())) };
            panic!("psdcascade: {}", msg.to_string_lossy());
        }
    }

    /// `Psd::new(fft, win)` with the reference's signature (src/psd.rs:137-152): `win` is ANY `Window<N>` -- the
    /// struct is public with public fields (src/psd.rs:12-20) -- and is uploaded as it is; a table equal to
    /// `Window::hann()` keeps the single-pass fused kernels, any other runs the generic two-pass kernels.  The plan
    /// `fft` is only held to the reference's assertion: the transform itself is the library's own hand-written FFT of
    /// the same length (an unnormalised forward DFT is an unnormalised forward DFT).
    pub fn new(fft: Arc<dyn Fft<f32>>, win: Arc<Window<N>>) -> Self {
        const { assert!(N >= 2) } // Nyquist and DC distinction (src/psd.rs:138)
        assert_eq!(N, fft.len()); // FFT and decimation block size compatibility (src/psd.rs:139)
        Self::with_device(win, PSDC_DEVICE_DEFAULT)
    }

    /// the same on a given HIP device
    pub fn with_device(win: Arc<Window<N>>, device: i32) -> Self {
        let s = unsafe {
            psdc_stage_create_window(N as u32, win.win.as_ptr(), win.power, win.nenbw, win.overlap, device as c_int)
        };
        let s = NonNull::new(s).unwrap_or_else(|| {
            let msg = unsafe { CStr::from_ptr(psdc_last_error(std::ptr::null())) };
            panic!("psdcascade: {}", msg.to_string_lossy())
        });
        Self { s, spectrum: vec![0.0; N / 2 + 1], buf: Vec::new() }
    }

    pub fn set_avg(&mut self, avg: u32) {
        self.check(unsafe { psdc_stage_set_avg(self.s.as_ptr(), avg) });
    }

    pub fn set_detrend(&mut self, d: Detrend) {
        self.check(unsafe { psdc_stage_set_detrend(self.s.as_ptr(), d as c_int) });
    }

    fn refresh(&mut self) {
        let s = self.s.as_ptr();
        self.check(unsafe { psdc_stage_get_spectrum(s, self.spectrum.as_mut_ptr()) });
        let mut len = 0usize;
        self.check(unsafe { psdc_stage_get_buf(s, std::ptr::null_mut(), 0, &mut len) });
        self.buf.resize(len, 0.0);
        if len > 0 {
            self.check(unsafe { psdc_stage_get_buf(s, self.buf.as_mut_ptr(), len, &mut len) });
        }
    }
}

impl<const N: usize> PsdStage for Psd<N> {
    fn process<'a>(&mut self, x: &[f32], y: &'a mut [f32]) -> &'a mut [f32] {
        let mut n = 0usize;
        // too small a `y` returns PSDC_ERR_CAPACITY -> panic, like the slice index at src/psd.rs:253
        self.check(unsafe { psdc_stage_process(self.s.as_ptr(), x.as_ptr(), x.len(), y.as_mut_ptr(), y.len(), &mut n) });
        self.refresh();
        &mut y[..n]
    }

    fn spectrum(&self) -> &[f32] {
        &self.spectrum
    }

    fn gain(&self) -> f32 {
        let mut g = 0f32;
        self.check(unsafe { psdc_stage_get_gain(self.s.as_ptr(), &mut g) });
        g
    }

    fn count(&self) -> u32 {
        let mut c = 0u32;
        self.check(unsafe { psdc_stage_get_count(self.s.as_ptr(), &mut c) });
        c
    }

    fn buf(&self) -> &[f32] {
        &self.buf
    }
}

impl<const N: usize> Clone for Psd<N> {
    fn clone(&self) -> Self {
        let s = unsafe { psdc_stage_clone(self.s.as_ptr()) };
        Self { s: NonNull::new(s).expect("psdc_stage_clone failed"), spectrum: self.spectrum.clone(), buf: self.buf.clone() }
    }
}

impl<const N: usize> Drop for Psd<N> {
    fn drop(&mut self) {
        unsafe { psdc_stage_destroy(self.s.as_ptr()) }
    }
}
