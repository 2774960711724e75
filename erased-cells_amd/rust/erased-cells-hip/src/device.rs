//! Device plumbing under the buffer types: initialisation, the calling thread's stream, HBM blocks.
use crate::error::{check, must, Result};
use crate::ffi::*;
use std::cell::Cell;
use std::os::raw::c_void;
use std::ptr;

/// Set up GPU `device` and make it the calling thread's device.  One process per GPU calls this once; a process
/// that drives several GPUs itself uses [`crate::sharded::ShardGroup`].
pub fn init(device: i32) -> Result<()> {
    check(unsafe { ec_init(device) })
}

thread_local! {
    static STREAM: Cell<ec_stream> = Cell::new(ptr::null_mut());
}

/// Route this thread's launches to `stream` (a `hipStream_t`; null = the default stream).
pub fn set_stream(stream: ec_stream) {
    STREAM.with(|s| s.set(stream));
}

/// The stream every call this thread makes is issued on.
pub fn stream() -> ec_stream {
    STREAM.with(|s| s.get())
}

/// One HBM allocation from the library's stream-ordered pool.  Operator results are allocated per call, as
/// the reference `collect()`s a fresh `Vec` per operator; hipMalloc/hipFree per operator would cost as much
/// as the kernel.
pub(crate) struct DeviceMem {
    ptr: *mut c_void,
    alloc_stream: ec_stream,
}

impl DeviceMem {
    pub(crate) fn new(bytes: usize) -> Self {
        let mut p = ptr::null_mut();
        let s = stream();
        if bytes > 0 {
            must(unsafe { ec_alloc_async(&mut p, bytes, s) }, "ec_alloc_async");
        }
        Self { ptr: p, alloc_stream: s }
    }
    pub(crate) fn ptr(&self) -> *mut c_void {
        self.ptr
    }
}

impl Drop for DeviceMem {
    /// Back to the pool on the ALLOCATING stream, ordered after everything queued so far on the stream that is
    /// current now (where the buffer's last operator ran if the thread switched streams in between).
    fn drop(&mut self) {
        if !self.ptr.is_null() {
            unsafe { ec_free_ordered(self.ptr, self.alloc_stream, stream()) };
        }
    }
}

/// Host -> HBM copy of `n` values.
pub(crate) fn upload<T: Copy>(dst: *mut c_void, src: &[T]) {
    if !src.is_empty() {
        must(unsafe { ec_upload(dst, src.as_ptr() as *const c_void, std::mem::size_of_val(src), stream()) }, "ec_upload");
    }
}

/// HBM -> host copy of `n` values of `T` (waits for the stream).
pub(crate) fn download<T: Copy + Default>(src: *const c_void, n: usize) -> Vec<T> {
    let mut v = vec![T::default(); n];
    if n > 0 {
        must(unsafe { ec_download(v.as_mut_ptr() as *mut c_void, src, n * std::mem::size_of::<T>(), stream()) }, "ec_download");
    }
    v
}
