// Run-time specialised figure-eight kernels (oh_jit.hip): one hiprtc module per kinematic chain, shared by the handles that use it.
#pragma once
#include <string>
#include <vector>

#include "oh_kernels.h"

struct FigSpec {
  hipModule_t mod = nullptr;
  hipFunction_t retract = nullptr, evalb = nullptr, evalb_zc = nullptr, tail = nullptr, tail_vel = nullptr, finalize = nullptr;
  int n = 0;               // chain length the kernels were instantiated for
  bool from_disk = false;  // code object came from the disk cache
  double seconds = 0.0;    // wall time of source generation + compilation (or cache read) + module load
};
std::string oh_jit_figure8_source(const oh_chain& chain, int N);
// hiprtc for gfx950 (needs no device); goes through the disk cache
int oh_jit_compile_cached(const std::string& src, std::vector<char>* code, bool* from_disk, std::string* err, bool ignore_disk = false);
// compile (or find) and load the kernels for this chain; *out stays owned by the process-wide cache
int oh_jit_figure8(const oh_chain& chain, int N, const FigSpec** out, std::string* err);
bool oh_jit_figure8_cached(const oh_chain& chain, int N);  // loaded in this process or present in the disk cache
hipError_t oh_spec_launch_eval(const FigSpec& sp, hipStream_t s, const FigParams& P, const FigBuffers& D, int slot, int part);
hipError_t oh_spec_launch_tail(const FigSpec& sp, hipStream_t s, const FigParams& P, const FigBuffers& D, int slot);
hipError_t oh_spec_launch_finalize(const FigSpec& sp, hipStream_t s, FigParams P, FigBuffers D, int only_done, double* f, double* kkt, int* iters, int* status);
hipError_t oh_spec_launch_tail_vel(const FigSpec& sp, hipStream_t s, FigParams P, FigBuffers D, GuardParams GP, GuardBuffers GB, int slot);
bool oh_spec_kernel_info(const FigSpec& sp, const char* name, OhKernelInfo* out);

// the library's disk cache of run-time compiled code objects ($OPTAS_HIP_CACHE), for its other generated kernels (oh_tape.hip)
bool oh_jit_disk_lookup(const std::string& key_text, const char* prefix, std::vector<char>* code);
void oh_jit_disk_store(const std::string& key_text, const char* prefix, const std::vector<char>& code);
void oh_jit_disk_drop(const std::string& key_text, const char* prefix);

// K1 (oh_fkjac_unit.h) for one chain: both layouts of the ABI
struct FkSpec {
  hipModule_t mod = nullptr;
  hipFunction_t soa = nullptr, aos = nullptr;
  int ndof = 0;  // sizes the staging tile of the reference-layout kernel
  bool from_disk = false;
  double seconds = 0.0;
};
std::string oh_jit_fkjac_source(const oh_chain& chain);
int oh_jit_fkjac(const oh_chain& chain, const FkSpec** out, std::string* err);
hipError_t oh_spec_launch_fk(const FkSpec& sp, hipStream_t s, bool soa, int n, const double* q, double* pose, double* J);
bool oh_spec_fk_kernel_info(const FkSpec& sp, OhKernelInfo* out);
