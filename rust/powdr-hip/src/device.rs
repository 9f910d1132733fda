//! Device memory for the HIP backend: what `openvm_cuda_common::d_buffer::DeviceBuffer` / `openvm_cuda_backend::base::
//! DeviceMatrix` are to the CUDA path (used at openvm/src/cuda_abi.rs:3-5,97-135 and
//! openvm/src/powdr_extension/trace_generator/cuda/mod.rs:264-270,331-332: `DeviceMatrix::with_capacity`, `.buffer()`,
//! `.fill_zero()`, `MemCopyH2D::to_device`, `DeviceMatrix::dummy()`), over hipMalloc/hipFree.
use crate::ffi;
use core::ffi::c_void;
use std::marker::PhantomData;

#[derive(Debug)]
pub struct HipError(pub i32);

impl HipError {
    /// the hipError_t (or library) code
    pub fn code(&self) -> i32 {
        self.0
    }
    /// `CudaError::from_result` of the reference wrappers (cuda_abi.rs:106-115): 0 = success.
    pub fn from_result(rc: i32) -> Result<(), HipError> {
        if rc == 0 {
            Ok(())
        } else {
            Err(HipError(rc))
        }
    }
}

pub struct DeviceBuffer<T> {
    ptr: *mut T,
    len: usize,
    _t: PhantomData<T>,
}

unsafe impl<T: Send> Send for DeviceBuffer<T> {}

impl<T: Copy> DeviceBuffer<T> {
    pub fn with_capacity(len: usize) -> Self {
        let mut p: *mut c_void = core::ptr::null_mut();
        let bytes = len.max(1) * core::mem::size_of::<T>();
        HipError::from_result(unsafe { ffi::hipMalloc(&mut p, bytes) }).expect("hipMalloc");
        Self { ptr: p as *mut T, len, _t: PhantomData }
    }
    pub fn as_ptr(&self) -> *const T {
        self.ptr
    }
    pub fn as_mut_ptr(&self) -> *mut T {
        self.ptr
    }
    pub fn len(&self) -> usize {
        self.len
    }
    pub fn is_empty(&self) -> bool {
        self.len == 0
    }
    pub fn fill_zero(&self) -> Result<(), HipError> {
        HipError::from_result(unsafe { ffi::hipMemset(self.ptr as *mut c_void, 0, self.len * core::mem::size_of::<T>()) })
    }
    pub fn to_host(&self) -> Result<Vec<T>, HipError> {
        let mut v = Vec::<T>::with_capacity(self.len);
        HipError::from_result(unsafe {
            ffi::hipMemcpy(v.as_mut_ptr() as *mut c_void, self.ptr as *const c_void, self.len * core::mem::size_of::<T>(),
                           ffi::HIP_MEMCPY_DEVICE_TO_HOST)
        })?;
        unsafe { v.set_len(self.len) };
        Ok(v)
    }
}

impl<T> Drop for DeviceBuffer<T> {
    fn drop(&mut self) {
        // hipFree synchronises with the device, so a buffer dropped right after an asynchronous launch (the reference
        // drops its table buffers that way) is safe.
        unsafe { ffi::hipFree(self.ptr as *mut c_void) };
    }
}

/// `MemCopyH2D::to_device` (cuda/mod.rs:331-332)
pub trait MemCopyH2D<T> {
    fn to_device(&self) -> Result<DeviceBuffer<T>, HipError>;
}
impl<T: Copy> MemCopyH2D<T> for [T] {
    fn to_device(&self) -> Result<DeviceBuffer<T>, HipError> {
        let b = DeviceBuffer::<T>::with_capacity(self.len());
        if !self.is_empty() {
            HipError::from_result(unsafe {
                ffi::hipMemcpy(b.as_mut_ptr() as *mut c_void, self.as_ptr() as *const c_void,
                               self.len() * core::mem::size_of::<T>(), ffi::HIP_MEMCPY_HOST_TO_DEVICE)
            })?;
        }
        Ok(b)
    }
}

/// Column-major device matrix, `idx = col * height + row` (apc_tracegen.cu:36,51).
pub struct DeviceMatrix<T> {
    buffer: DeviceBuffer<T>,
    height: usize,
    width: usize,
}

impl<T: Copy> DeviceMatrix<T> {
    pub fn with_capacity(height: usize, width: usize) -> Self {
        Self { buffer: DeviceBuffer::with_capacity(height * width), height, width }
    }
    /// The 0 x 0 matrix an uncalled APC reports (cuda/mod.rs:412-414 `DeviceMatrix::dummy`).
    pub fn dummy() -> Self {
        Self::with_capacity(0, 0)
    }
    pub fn buffer(&self) -> &DeviceBuffer<T> {
        &self.buffer
    }
    /// `MatrixDimensions` (cuda_abi.rs:5,103)
    pub fn height(&self) -> usize {
        self.height
    }
    pub fn width(&self) -> usize {
        self.width
    }
}
