//! gpu_ivf.rs — the rabitq-rs side of the `librbq.so` boundary (include/rbq.h).
//!
//! Drop this file into the crate as `src/gpu_ivf.rs` (`pub mod gpu_ivf;` behind a `gpu` feature) together with
//! `integration/build.rs`.  It gives an `IvfRabitqIndex`-shaped handle whose `search`, `search_filtered` and
//! `batch_search` run on the MI355X and return what the CPU path returns (reference src/ivf.rs:1705-1752).
//! Binding A of INTEGRATION.md: the index crosses the boundary as the RBQ1-v3 byte stream the unmodified crate
//! writes (`IvfRabitqIndex::save_to_writer`, src/ivf.rs:1317-1474).
//!
//! No Rust toolchain exists in the build container of this repository, so this file has never been compiled
//! there; `tests/test_rust_shim.py` checks every `extern "C"` signature, every `#[repr(C)]` struct and the
//! error-code table below against include/rbq.h mechanically.

use std::os::raw::{c_char, c_int, c_void};

use roaring::RoaringBitmap;

use crate::ivf::{IvfRabitqIndex, SearchParams, SearchResult};
use crate::RabitqError;

// ---- include/rbq.h -------------------------------------------------------------------------------------------------

#[repr(C)]
pub struct RbqIndex {
    _p: [u8; 0],
}
#[repr(C)]
pub struct RbqBuilder {
    _p: [u8; 0],
}

/// `rbq_header`
#[repr(C)]
pub struct RbqHeader {
    pub dim: u32,
    pub padded_dim: u32,
    pub metric: u8,
    pub rotator: u8,
    pub ex_bits: u8,
    pub reserved: u8,
    pub n_vectors: u64,
    pub n_lists: u64,
    pub rotator_blob: *const u8,
    pub rotator_len: u64,
}

/// `rbq_list_view`: one `ClusterData` (src/ivf.rs:205-242), borrowed for the call
#[repr(C)]
pub struct RbqListView {
    pub centroid: *const f32,
    pub n: u64,
    pub ids: *const u64,
    pub batch_data: *const u8,
    pub batch_len: u64,
    pub ex_codes: *const u8,
    pub f_add_ex: *const f32,
    pub f_rescale_ex: *const f32,
}

/// `rbq_diag` == `SearchDiagnostics` (src/ivf.rs:150-155)
#[repr(C)]
#[derive(Default, Clone, Copy, Debug)]
pub struct RbqDiag {
    pub estimated: u64,
    pub skipped_by_lower_bound: u64,
    pub extended_evaluations: u64,
}

pub const RBQ_OK: c_int = 0;
pub const RBQ_DIMENSION_MISMATCH: c_int = 1;
pub const RBQ_INVALID_CONFIG: c_int = 2;
pub const RBQ_EMPTY_INDEX: c_int = 3;
pub const RBQ_IO: c_int = 4;
pub const RBQ_INVALID_PERSISTENCE: c_int = 5;
pub const RBQ_DEVICE: c_int = 6;

extern "C" {
    fn rbq_index_create(hdr: *const RbqHeader, lists: *const RbqListView, n_devices: c_int, devices: *const c_int,
                        out: *mut *mut RbqIndex) -> c_int;
    fn rbq_index_load_rbq1(bytes: *const c_void, len: usize, n_devices: c_int, devices: *const c_int,
                           out: *mut *mut RbqIndex) -> c_int;
    fn rbq_index_build_device(hdr: *const RbqHeader, centroids: *const f32, d_data: *const f32, d_assign: *const u32,
                              n: u64, t_const: f32, device: c_int, out: *mut *mut RbqIndex) -> c_int;
    fn rbq_build_stream_begin(hdr: *const RbqHeader, centroids: *const f32, list_sizes: *const u32, t_const: f32,
                              device: c_int, out: *mut *mut RbqBuilder) -> c_int;
    fn rbq_build_stream_push(b: *mut RbqBuilder, vectors: *const f32, assign: *const u32, first_id: u64, count: u64) -> c_int;
    fn rbq_build_stream_finish(b: *mut RbqBuilder, n_devices: c_int, devices: *const c_int, out: *mut *mut RbqIndex) -> c_int;
    fn rbq_build_stream_abort(b: *mut RbqBuilder);
    fn rbq_index_destroy(idx: *mut RbqIndex);
    fn rbq_index_len(idx: *const RbqIndex) -> u64;
    fn rbq_index_cluster_count(idx: *const RbqIndex) -> u64;
    fn rbq_index_dim(idx: *const RbqIndex) -> u32;
    fn rbq_index_padded_dim(idx: *const RbqIndex) -> u32;
    fn rbq_index_device_count(idx: *const RbqIndex) -> u32;
    fn rbq_search_batch(idx: *const RbqIndex, queries: *const f32, nq: u64, query_dim: u32, top_k: u32, nprobe: u32,
                        filter_words: *const u32, filter_nbits: u64, out_ids: *mut u64, out_scores: *mut f32,
                        out_counts: *mut u32, diag: *mut RbqDiag) -> c_int;
    fn rbq_posting_scan_batch(idx: *const RbqIndex, queries: *const f32, nq: u64, query_dim: u32, top_k: u32,
                              list_ids: *const u32, list_counts: *const u32, max_lists: u32, out_ids: *mut u64,
                              out_scores: *mut f32, out_counts: *mut u32) -> c_int;
    fn rbq_search_batch_device(idx: *const RbqIndex, d_queries: *const f32, nq: u64, query_dim: u32, top_k: u32,
                               nprobe: u32, d_filter_words: *const u32, filter_nbits: u64, d_out_ids: *mut u64,
                               d_out_scores: *mut f32, d_out_counts: *mut u32, d_diag: *mut RbqDiag,
                               hip_stream: *mut c_void) -> c_int;
    fn rbq_release_stream(idx: *mut RbqIndex, hip_stream: *mut c_void) -> c_int;
    fn rbq_host_alloc(bytes: usize) -> *mut c_void;
    fn rbq_host_free(p: *mut c_void);
    fn rbq_index_set_rerank_vectors(idx: *mut RbqIndex, vectors: *const f32, n: u64) -> c_int;
    fn rbq_debug_set_option(idx: *mut RbqIndex, name: *const c_char, value: c_int) -> c_int;
    fn rbq_strerror(code: c_int) -> *const c_char;
    fn rbq_last_error_detail(buf: *mut c_char, n: usize) -> c_int;
    fn rbq_abi_version() -> u32;
}

// ---- errors: RabitqError (src/lib.rs:39-57) <- rbq codes --------------------------------------------------------------

/// `InvalidConfig` / `InvalidPersistence` carry a `&'static str`: the detail strings the library can return for
/// an index written by the crate are the crate's own (src/ivf.rs:1484-1702), listed here so that no allocation is
/// leaked for them; anything else (a device message) is leaked once per distinct occurrence.
const KNOWN_DETAILS: &[&str] = &[
    "unrecognized file header",
    "unsupported index format version (expected V3 with unified memory layout)",
    "dimension must be positive",
    "padded_dim must be >= dim",
    "unknown metric tag",
    "unknown rotator type tag",
    "ex_bits out of range",
    "total_bits out of range",
    "total_bits does not match ex_bits",
    "FHT rotator flip bits length mismatch",
    "rotator matrix length mismatch",
    "cluster size exceeds reasonable limits - possible corruption",
    "batch_data length mismatch - possible corruption or version incompatibility",
    "ex_code_packed length mismatch - possible corruption or version incompatibility",
    "vector count metadata mismatch",
    "checksum mismatch",
    "Unsupported ex_bits: only 0 (1-bit total), 2 (3-bit total), and 6 (7-bit total) are supported",
    "Dimension must be multiple of 16 for SIMD",
    "FHT rotator requires dimension to be multiple of 64",
    "padded_dim > 2048 (high-accuracy i32 LUT mode) is not supported",
    "nlist must be positive",
    "null index",
    "null buffer",
];

fn static_detail(detail: String) -> &'static str {
    for k in KNOWN_DETAILS {
        if *k == detail {
            return k;
        }
    }
    Box::leak(detail.into_boxed_str())
}

fn last_detail() -> String {
    let mut buf = vec![0u8; 256];
    let n = unsafe { rbq_last_error_detail(buf.as_mut_ptr() as *mut c_char, buf.len()) };
    let n = (n.max(0) as usize).min(buf.len() - 1);
    String::from_utf8_lossy(&buf[..n]).into_owned()
}

/// One arm per code of include/rbq.h.  `expected` / `got` are the dimensions of the call (the library reports
/// them in its detail string as "expected E, got G"; the caller knows them).
fn map_err(rc: c_int, expected: usize, got: usize) -> Result<(), RabitqError> {
    match rc {
        RBQ_OK => Ok(()),
        RBQ_DIMENSION_MISMATCH => Err(RabitqError::DimensionMismatch { expected, got }),
        RBQ_INVALID_CONFIG => Err(RabitqError::InvalidConfig(static_detail(last_detail()))),
        RBQ_EMPTY_INDEX => Err(RabitqError::EmptyIndex),
        RBQ_IO => Err(RabitqError::Io(std::io::Error::new(std::io::ErrorKind::Other, last_detail()))),
        RBQ_INVALID_PERSISTENCE => Err(RabitqError::InvalidPersistence(static_detail(last_detail()))),
        RBQ_DEVICE => Err(RabitqError::Io(std::io::Error::new(
            std::io::ErrorKind::Other,
            format!("device: {}", last_detail()),
        ))),
        other => Err(RabitqError::Io(std::io::Error::new(
            std::io::ErrorKind::Other,
            format!("librbq: unknown error code {other}"),
        ))),
    }
}

// ---- the handle --------------------------------------------------------------------------------------------------------

/// A GPU-resident copy of an `IvfRabitqIndex` (replicated on one or more devices).
pub struct GpuIvf {
    h: *mut RbqIndex,
    dim: usize,
    /// 1 + the largest id of the index when it is known (an index built by the crate stores ids 0..len, src/ivf.rs:1181,1192-1198):
    /// filter bits at or beyond it can never match a vector, so the dense bitset of `search_filtered` stops there.
    id_bound: Option<u64>,
}

// rbq_search_batch is re-entrant on one handle (include/rbq.h); the reference's search is `&self` and is called from
// Rayon workers (src/ivf.rs:1748-1751).
unsafe impl Send for GpuIvf {}
unsafe impl Sync for GpuIvf {}

impl GpuIvf {
    /// `index.save_to_writer(&mut buf)` (src/ivf.rs:1317) -> one replica on the current device.
    pub fn from_index(index: &IvfRabitqIndex) -> Result<Self, RabitqError> {
        Self::from_index_on(index, &[])
    }

    /// The same on the given HIP device ordinals (one replica each; `rbq_search_batch` shards a batch over them).
    pub fn from_index_on(index: &IvfRabitqIndex, devices: &[i32]) -> Result<Self, RabitqError> {
        let mut buf = Vec::new();
        index.save_to_writer(&mut buf)?;
        let mut g = Self::from_rbq1_bytes(&buf, devices)?;
        g.id_bound = Some(index.len() as u64); // ids of a crate-built index are the vector indexes 0..len
        Ok(g)
    }

    /// An RBQ1-v3 byte stream as written by `save_to_writer` / `save_to_path`; validated like `load_from_reader`
    /// (src/ivf.rs:1484-1702), CRC included.
    pub fn from_rbq1_bytes(bytes: &[u8], devices: &[i32]) -> Result<Self, RabitqError> {
        let mut h: *mut RbqIndex = std::ptr::null_mut();
        let devs: Vec<c_int> = devices.iter().map(|d| *d as c_int).collect();
        let (n, p) = if devs.is_empty() { (1, std::ptr::null()) } else { (devs.len() as c_int, devs.as_ptr()) };
        let rc = unsafe { rbq_index_load_rbq1(bytes.as_ptr() as *const c_void, bytes.len(), n, p, &mut h) };
        map_err(rc, 0, 0)?;
        let dim = unsafe { rbq_index_dim(h) } as usize;
        Ok(Self { h, dim, id_bound: None }) // (a stream of unknown origin may hold any u64 ids)
    }

    pub fn len(&self) -> usize {
        unsafe { rbq_index_len(self.h) as usize }
    }
    pub fn is_empty(&self) -> bool {
        self.len() == 0
    }
    pub fn cluster_count(&self) -> usize {
        unsafe { rbq_index_cluster_count(self.h) as usize }
    }
    pub fn device_count(&self) -> usize {
        unsafe { rbq_index_device_count(self.h) as usize }
    }

    /// `IvfRabitqIndex::search` (src/ivf.rs:1705-1711).
    pub fn search(&self, query: &[f32], params: SearchParams) -> Result<Vec<SearchResult>, RabitqError> {
        self.search_one(query, params, None, None)
    }

    /// `IvfRabitqIndex::search_filtered` (src/ivf.rs:1723-1730): the scan tests `filter.contains(id as u32)`
    /// (src/ivf.rs:2018-2022); the bitmap crosses the boundary as a dense bitset over the u32 id space.
    pub fn search_filtered(
        &self,
        query: &[f32],
        params: SearchParams,
        filter: &RoaringBitmap,
    ) -> Result<Vec<SearchResult>, RabitqError> {
        let (words, nbits) = dense_words(filter, self.id_bound);
        self.search_one(query, params, Some((&words, nbits)), None)
    }

    /// `search` with the reference's `SearchDiagnostics` counters (src/ivf.rs:150-155).
    pub fn search_with_diagnostics(
        &self,
        query: &[f32],
        params: SearchParams,
    ) -> Result<(Vec<SearchResult>, RbqDiag), RabitqError> {
        let mut d = RbqDiag::default();
        let r = self.search_one(query, params, None, Some(&mut d))?;
        Ok((r, d))
    }

    fn search_one(
        &self,
        query: &[f32],
        params: SearchParams,
        filter: Option<(&[u32], u64)>,
        diag: Option<&mut RbqDiag>,
    ) -> Result<Vec<SearchResult>, RabitqError> {
        let k = params.top_k;
        let mut ids = vec![0u64; k.max(1)];
        let mut sc = vec![0f32; k.max(1)];
        let mut cnt = 0u32;
        let (fw, fb) = match filter {
            Some((w, n)) => (w.as_ptr(), n),
            None => (std::ptr::null(), 0u64),
        };
        let dp = match diag {
            Some(d) => d as *mut RbqDiag,
            None => std::ptr::null_mut(),
        };
        // the query length travels as query_dim: EmptyIndex is reported before DimensionMismatch, as
        // search_fastscan does (src/ivf.rs:1761-1769)
        let rc = unsafe {
            rbq_search_batch(self.h, query.as_ptr(), 1, query.len() as u32, k as u32, params.nprobe as u32, fw, fb,
                             ids.as_mut_ptr(), sc.as_mut_ptr(), &mut cnt, dp)
        };
        map_err(rc, self.dim, query.len())?;
        Ok((0..cnt as usize).map(|i| SearchResult { id: ids[i] as usize, score: sc[i] }).collect())
    }

    /// `IvfRabitqIndex::batch_search` (src/ivf.rs:1743-1752): per-query results in input order.  Queries of the
    /// right length go to the GPU in ONE call; a query of the wrong length gets its own `DimensionMismatch`
    /// (or `EmptyIndex`, which the reference checks first) without disturbing the others.
    pub fn batch_search(&self, queries: &[&[f32]], params: SearchParams) -> Vec<Result<Vec<SearchResult>, RabitqError>> {
        let k = params.top_k;
        let good: Vec<usize> = (0..queries.len()).filter(|&i| queries[i].len() == self.dim).collect();
        let mut out: Vec<Option<Result<Vec<SearchResult>, RabitqError>>> = (0..queries.len()).map(|_| None).collect();
        if !good.is_empty() {
            let nq = good.len();
            // The queries are gathered into ONE flat buffer anyway (the reference takes &[&[f32]]): gather them straight into
            // page-locked memory (a per-thread scratch, grown on demand and reused) — the library then reads them in place and
            // writes the results in place, no staging copy on either side (3.6 M against 2.6 M queries/s
            // for one 1024-query call on the GIST-1M shape).
            let rc_and_rows: Result<Vec<Vec<SearchResult>>, c_int> = SCRATCH.with(|cell| {
                let mut sc = cell.borrow_mut();
                if !sc.ensure(nq * self.dim, (nq * k).max(1), nq) {
                    return Err(RBQ_IO);
                }
                let s = &mut *sc;
                let (fq, fi, fs, fc) = (s.q.as_mut().unwrap(), s.ids.as_mut().unwrap(), s.sc.as_mut().unwrap(), s.cnt.as_mut().unwrap());
                {
                    let flat = fq.as_mut_slice();
                    for (j, &i) in good.iter().enumerate() {
                        flat[j * self.dim..(j + 1) * self.dim].copy_from_slice(queries[i]);
                    }
                }
                let rc = unsafe {
                    rbq_search_batch(self.h, fq.as_slice().as_ptr(), nq as u64, self.dim as u32, k as u32, params.nprobe as u32,
                                     std::ptr::null(), 0, fi.as_mut_slice().as_mut_ptr(), fs.as_mut_slice().as_mut_ptr(),
                                     fc.as_mut_slice().as_mut_ptr(), std::ptr::null_mut())
                };
                if rc != RBQ_OK {
                    return Err(rc);
                }
                let (ids, scs, cnt) = (fi.as_slice(), fs.as_slice(), fc.as_slice());
                Ok((0..nq)
                    .map(|j| (0..cnt[j] as usize).map(|r| SearchResult { id: ids[j * k + r] as usize, score: scs[j * k + r] }).collect())
                    .collect())
            });
            match rc_and_rows {
                Ok(rows) => {
                    for (row, &i) in rows.into_iter().zip(good.iter()) {
                        out[i] = Some(Ok(row));
                    }
                }
                Err(rc) => {
                    for &i in &good {
                        out[i] = Some(match map_err(rc, self.dim, self.dim) {
                            Ok(()) => Ok(Vec::new()),
                            Err(e) => Err(e),
                        });
                    }
                }
            }
        }
        for (i, slot) in out.iter_mut().enumerate() {
            if slot.is_none() {
                *slot = Some(self.search(queries[i], params)); // wrong length: the library's own error order
            }
        }
        out.into_iter().map(|r| r.unwrap()).collect()
    }

    /// Diagnostic switches of the library (`rbq_debug_set_option`), e.g. ("lazy_select", 0).
    pub fn set_option(&self, name: &str, value: i32) -> Result<(), RabitqError> {
        let c = std::ffi::CString::new(name).map_err(|_| RabitqError::InvalidConfig("option name contains NUL"))?;
        map_err(unsafe { rbq_debug_set_option(self.h, c.as_ptr(), value as c_int) }, 0, 0)
    }

    /// Library/ABI version (major << 16 | minor).
    pub fn abi_version() -> u32 {
        unsafe { rbq_abi_version() }
    }
}

impl Drop for GpuIvf {
    fn drop(&mut self) {
        unsafe { rbq_index_destroy(self.h) }
    }
}

/// Per-thread page-locked scratch of `batch_search` (queries in, ids / scores / counts out), grown on demand.
struct Scratch {
    q: Option<PinnedBuf<f32>>,
    ids: Option<PinnedBuf<u64>>,
    sc: Option<PinnedBuf<f32>>,
    cnt: Option<PinnedBuf<u32>>,
}
impl Scratch {
    /// Page-locked memory is a scarce, per-thread resource here (one scratch per Rayon worker): a buffer that a single large batch
    /// once blew up is given back when a call needs less than a quarter of it (and it holds more than 16 MB).
    fn grow<T: Copy>(slot: &mut Option<PinnedBuf<T>>, len: usize) -> bool {
        let have = slot.as_ref().map_or(0, |b| b.as_slice().len());
        let oversized = have > 4 * len.max(1) && have * std::mem::size_of::<T>() > (16 << 20);
        if have >= len && !oversized {
            return true;
        }
        *slot = PinnedBuf::new(len + len / 4);
        slot.is_some()
    }
    fn ensure(&mut self, nq_dim: usize, nres: usize, nq: usize) -> bool {
        Self::grow(&mut self.q, nq_dim) && Self::grow(&mut self.ids, nres) && Self::grow(&mut self.sc, nres) && Self::grow(&mut self.cnt, nq)
    }
}
thread_local! {
    static SCRATCH: std::cell::RefCell<Scratch> = std::cell::RefCell::new(Scratch { q: None, ids: None, sc: None, cnt: None });
}

/// RoaringBitmap -> dense little-endian words: bit i of the bitset set <=> `filter.contains(i)`.
/// `id_bound`: bits at or beyond it are dropped (they cannot match an indexed vector) — a filter that happens to contain one
/// high u32 id would otherwise cost 512 MB of zeros per call.
fn dense_words(filter: &RoaringBitmap, id_bound: Option<u64>) -> (Vec<u32>, u64) {
    let mut nbits = match filter.max() {
        Some(m) => m as u64 + 1,
        None => 0,
    };
    if let Some(b) = id_bound {
        nbits = nbits.min(b);
    }
    let mut words = vec![0u32; ((nbits + 31) / 32) as usize];
    for id in filter.iter() {
        if (id as u64) >= nbits {
            break; // (ascending iteration)
        }
        words[(id >> 5) as usize] |= 1u32 << (id & 31);
    }
    (words, nbits)
}

/// Page-locked host buffer (`rbq_host_alloc`): `rbq_search_batch` reads queries from / writes results to such
/// buffers in place (no staging copy).
pub struct PinnedBuf<T: Copy> {
    p: *mut T,
    len: usize,
}
impl<T: Copy> PinnedBuf<T> {
    pub fn new(len: usize) -> Option<Self> {
        let p = unsafe { rbq_host_alloc(len.max(1) * std::mem::size_of::<T>()) } as *mut T;
        if p.is_null() {
            None
        } else {
            Some(Self { p, len })
        }
    }
    pub fn as_mut_slice(&mut self) -> &mut [T] {
        unsafe { std::slice::from_raw_parts_mut(self.p, self.len) }
    }
    pub fn as_slice(&self) -> &[T] {
        unsafe { std::slice::from_raw_parts(self.p, self.len) }
    }
}
impl<T: Copy> Drop for PinnedBuf<T> {
    fn drop(&mut self) {
        unsafe { rbq_host_free(self.p as *mut c_void) }
    }
}

// The remaining imports of the extern block (in-crate ClusterData binding, GPU encoder, MSTG scan, device entry) are
// used by binding B / E / F of INTEGRATION.md; they are declared here so that ONE file carries the whole boundary.
#[allow(dead_code)]
fn _boundary_is_complete() {
    let _ = (
        rbq_index_create as usize,
        rbq_index_build_device as usize,
        rbq_build_stream_begin as usize,
        rbq_build_stream_push as usize,
        rbq_build_stream_finish as usize,
        rbq_build_stream_abort as usize,
        rbq_index_padded_dim as usize,
        rbq_posting_scan_batch as usize,
        rbq_search_batch_device as usize,
        rbq_release_stream as usize,
        rbq_index_set_rerank_vectors as usize,
        rbq_strerror as usize,
    );
}
