//! N-GPU encode / decode of ONE member: one rank per GPU, the library's own drivers
//! (`lfx_sharded_encode_begin` / `_finish`, `lfx_sharded_decode`, include/lfx.h) with the caller's collectives.
//!
//! The reference has one encoder and one running checksum (`gzip::Encoder::finish`, src/gzip.rs:858-868;
//! src/checksum.rs:22-33); here every rank encodes its own whole blocks, the ranks exchange 32 bytes each
//! (bit length, byte count, CRC-32, Adler-32), emit at their global bit offset, and the shards travel once to rank 0.
//! The collectives are a [`Collective`] the caller implements over whatever connects its ranks (MPI, a socket mesh, ...),
//! or RCCL over xGMI through [`Comm::rccl`].  Buffers are DEVICE pointers (the caller's allocator: hipMalloc, a torch
//! tensor, ...): this crate allocates nothing on the device.
use std::io;
use std::os::raw::{c_int, c_void};

use crate::{ffi, Context};

/// What the drivers need between ranks.  Every rank calls the same sequence of collectives.
pub trait Collective {
    /// every rank contributes `send`; `recv` (world x send.len(), rank order) is complete on return
    fn allgather(&self, send: &[u8], recv: &mut [u8]) -> io::Result<()>;
    /// post a transfer of `bytes` bytes of DEVICE memory (may return before it completes; may only collect it)
    fn isend(&self, d_buf: *const c_void, bytes: u64, to_rank: u32) -> io::Result<()>;
    fn irecv(&self, d_buf: *mut c_void, bytes: u64, from_rank: u32) -> io::Result<()>;
    /// everything posted so far begins to move now; returns without waiting (`encode_begin` calls it last, so the shards
    /// travel while the caller works).  The default suits a collective whose `isend` / `irecv` start at once.
    fn start(&self) -> io::Result<()> { Ok(()) }
    /// everything this rank posted is complete
    fn wait(&self) -> io::Result<()>;
}

/// An `lfx_comm` over a [`Collective`] (or over RCCL).  The collective must outlive it.
pub struct Comm<'a> { raw: ffi::lfx_comm, rccl: bool, _c: std::marker::PhantomData<&'a ()> }

extern "C" fn tr_allgather<C: Collective>(user: *mut c_void, send: *const c_void, recv: *mut c_void, bytes: u64) -> c_int {
    let (c, w) = unsafe { &*(user as *const (&C, u32)) };
    let s = unsafe { std::slice::from_raw_parts(send as *const u8, bytes as usize) };
    let r = unsafe { std::slice::from_raw_parts_mut(recv as *mut u8, bytes as usize * *w as usize) };
    c.allgather(s, r).is_err() as c_int
}
extern "C" fn tr_isend<C: Collective>(user: *mut c_void, d: *const c_void, bytes: u64, to: u32) -> c_int {
    let (c, _) = unsafe { &*(user as *const (&C, u32)) };
    c.isend(d, bytes, to).is_err() as c_int
}
extern "C" fn tr_irecv<C: Collective>(user: *mut c_void, d: *mut c_void, bytes: u64, from: u32) -> c_int {
    let (c, _) = unsafe { &*(user as *const (&C, u32)) };
    c.irecv(d, bytes, from).is_err() as c_int
}
extern "C" fn tr_wait<C: Collective>(user: *mut c_void) -> c_int {
    let (c, _) = unsafe { &*(user as *const (&C, u32)) };
    c.wait().is_err() as c_int
}
extern "C" fn tr_start<C: Collective>(user: *mut c_void) -> c_int {
    let (c, _) = unsafe { &*(user as *const (&C, u32)) };
    c.start().is_err() as c_int
}

impl<'a> Comm<'a> {
    /// `slot` keeps the (collective, world) pair the callbacks read: it must live as long as the `Comm`
    pub fn new<C: Collective>(slot: &'a mut Option<(&'a C, u32)>, c: &'a C, rank: u32, world: u32) -> Comm<'a> {
        *slot = Some((c, world));
        let user = slot.as_ref().unwrap() as *const (&C, u32) as *mut c_void;
        Comm { raw: ffi::lfx_comm { user, rank, world, allgather: Some(tr_allgather::<C>), isend: Some(tr_isend::<C>),
                                    irecv: Some(tr_irecv::<C>), wait: Some(tr_wait::<C>), start: Some(tr_start::<C>) }, rccl: false, _c: std::marker::PhantomData }
    }
    /// one rank, no collective at all (world = 1)
    pub fn single() -> Comm<'static> {
        Comm { raw: ffi::lfx_comm { user: std::ptr::null_mut(), rank: 0, world: 1, allgather: None, isend: None, irecv: None, wait: None, start: None },
               rccl: false, _c: std::marker::PhantomData }
    }
    /// RCCL over xGMI: `nccl_comm` is an `ncclComm_t`, `hip_stream` the `hipStream_t` its collectives run on
    /// (librccl is loaded at run time).
    pub fn rccl(nccl_comm: *mut c_void, hip_stream: *mut c_void, rank: u32, world: u32) -> io::Result<Comm<'static>> {
        let mut raw = ffi::lfx_comm { user: std::ptr::null_mut(), rank, world, allgather: None, isend: None, irecv: None, wait: None, start: None };
        let rc = unsafe { ffi::lfx_comm_rccl(nccl_comm, hip_stream, rank, world, &mut raw) };
        if rc != ffi::LFX_OK { return Err(io::Error::new(io::ErrorKind::Other, format!("RCCL is not available (status {})", rc))); }
        Ok(Comm { raw, rccl: true, _c: std::marker::PhantomData })
    }
}
impl<'a> Drop for Comm<'a> {
    fn drop(&mut self) { if self.rccl { unsafe { ffi::lfx_comm_rccl_free(&mut self.raw) } } }
}

fn err(ctx: &Context, rc: c_int, what: &str) -> io::Error {
    io::Error::new(io::ErrorKind::Other, format!("{} failed (status {}): {}", what, rc, ctx.last_error()))
}

/// A sharded encode in flight: the shards are travelling to rank 0; `finish` completes the member there.
pub struct EncodeInFlight<'a> { ctx: &'a Context, comm: &'a Comm<'a>, state: *mut ffi::lfx_sharded_enc, pub part: ffi::lfx_sharded_part }

/// `gzip::Encoder` / `zlib::Encoder` / `deflate::Encoder` over N GPUs: this rank's slice `d_in[..n]` of the input (whole blocks
/// on every rank but the last) → its shard in `d_part`; rank 0 also passes the member buffer and a staging buffer for the other
/// ranks' shards.  `write_size`: the write schedule (0 = one `write_all`, 8192 = `io::copy`'s).
#[allow(clippy::too_many_arguments)]
pub fn encode_begin<'a>(ctx: &'a Context, comm: &'a Comm<'a>, format: c_int, opts: &ffi::lfx_encode_opts, write_size: u64,
                        d_in: *const c_void, n: u64, d_part: *mut c_void, part_cap: u64, d_member: *mut c_void, member_cap: u64,
                        d_staging: *mut c_void, staging_cap: u64) -> io::Result<EncodeInFlight<'a>> {
    let sched = ffi::lfx_schedule { kind: if write_size == 0 { 0 } else { 1 }, fixed_write: write_size, writes: std::ptr::null(), n_writes: 0 };
    let mut state: *mut ffi::lfx_sharded_enc = std::ptr::null_mut();
    let mut part = ffi::lfx_sharded_part::default();
    let rc = unsafe { ffi::lfx_sharded_encode_begin(ctx.0, &comm.raw, format, opts, &sched, d_in, n, d_part, part_cap, d_member, member_cap,
                                                    d_staging, staging_cap, &mut state, &mut part) };
    if rc != ffi::LFX_OK { return Err(err(ctx, rc, "lfx_sharded_encode_begin")); }
    Ok(EncodeInFlight { ctx, comm, state, part })
}
impl<'a> EncodeInFlight<'a> {
    /// → the member's length on rank 0 (0 elsewhere)
    pub fn finish(mut self) -> io::Result<u64> {
        let mut len = 0u64;
        let state = std::mem::replace(&mut self.state, std::ptr::null_mut());      // (finish frees it: Drop must not)
        let rc = unsafe { ffi::lfx_sharded_encode_finish(self.ctx.0, &self.comm.raw, state, &mut len) };
        if rc != ffi::LFX_OK { return Err(err(self.ctx, rc, "lfx_sharded_encode_finish")); }
        Ok(len)
    }
}
/// Dropped without `finish` (an early return with `?`): the transfers this rank posted are still completed and the state is
/// freed — the peers must not be left waiting in their own `finish`.
impl<'a> Drop for EncodeInFlight<'a> {
    fn drop(&mut self) {
        if !self.state.is_null() {
            let mut len = 0u64;
            unsafe { ffi::lfx_sharded_encode_finish(self.ctx.0, &self.comm.raw, self.state, &mut len) };
            self.state = std::ptr::null_mut();
        }
    }
}

/// the compressed byte range rank `rank` of `world` scans: (lo, hi, hold_hi) — it must hold the member's bytes [lo, hold_hi)
pub fn byte_range(first_byte: u64, member_len: u64, rank: u32, world: u32) -> (u64, u64, u64) {
    let (mut lo, mut hi, mut hold) = (0u64, 0u64, 0u64);
    unsafe { ffi::lfx_sharded_byte_range(first_byte, member_len, rank, world, &mut lo, &mut hi, &mut hold) };
    (lo, hi, hold)
}

/// `gzip::Decoder` over N GPUs for ONE member cut by compressed bytes: this rank's slice of the output in `d_out`, and the
/// checksums of the whole output (compare with the trailer).  `first_bit`: the member's first DEFLATE bit.
#[allow(clippy::too_many_arguments)]
pub fn decode(ctx: &Context, comm: &Comm, d_part: *const c_void, n_part: u64, lo_byte: u64, hi_byte: u64, first_bit: u64, member_len: u64,
              d_out: *mut c_void, cap: u64) -> io::Result<ffi::lfx_sharded_slice> {
    let mut sl = ffi::lfx_sharded_slice::default();
    let rc = unsafe { ffi::lfx_sharded_decode(ctx.0, &comm.raw, d_part, n_part, lo_byte, hi_byte, first_bit, member_len, d_out, cap, &mut sl) };
    if rc != ffi::LFX_OK { return Err(err(ctx, rc, "lfx_sharded_decode")); }
    Ok(sl)
}
