// gfx950 side of the platform hooks oh_figure8_units.h asks for (see there).
#pragma once
#include "oh_device.h"

// Row addressing for the hot loops: a stage array is [row = t*K + k][Bp].  Through a buffer resource the row offset travels in an SGPR
// and the lane adds ONE 32-bit byte offset shared by every stream ("buffer_load_dwordx2 v, v_off, s[rsrc], s_row offen"); with flat
// global pointers the compiler keeps a 64-bit VGPR address per stream alive (k_step: 50 of them, 55 registers spilled inside its
// serial sweep).  A RowBuf is rebased per knot (scalar ALU), so the SGPR offset k*Bp*8 always fits 32 bits.
struct RowBuf {
  __amdgpu_buffer_rsrc_t r;
};
OH_DEV RowBuf rowbuf(const double* knot_base) {
  return RowBuf{__builtin_amdgcn_make_buffer_rsrc((void*)knot_base, 0, 0xFFFFFFFF, 0x00020000)};  // raw buffer, gfx9 data format word
}
OH_DEV double rb_ld(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes) {
  typedef int v2i __attribute__((ext_vector_type(2)));
  const v2i v = __builtin_amdgcn_raw_buffer_load_b64(rb.r, lane_bytes, row_bytes, 0);
  return __builtin_bit_cast(double, v);
}
OH_DEV void rb_st(const RowBuf& rb, const unsigned row_bytes, const unsigned lane_bytes, const double x) {
  typedef int v2i __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, x), rb.r, lane_bytes, row_bytes, 0);
}
OH_DEV void oh_count(unsigned long long* c) { atomicAdd(c, 1ULL); }
OH_DEV int oh_take_ticket(int* c) { return atomicAdd(c, 1); }  // a slot of a device-side list
// the compiler sinks loads to their first use, i.e. below the early-exit branches: an empty asm that takes the value as a VGPR operand keeps the
// whole batch of loads issued before it above the branch (one s_waitcnt for all of them)
OH_DEV void oh_fence(const double x) { asm volatile("" ::"v"(x)); }
