// Kernel-selection / tuning switches of the launchers as EXPLICIT library state (include/bmv.h: bmv_tuning_set / _get /
// _clear / _name) instead of environment variables read -- and cached -- inside the C launchers (round 3): visible in the
// header, settable and changeable at any time, never dependent on the process environment.  The Python host applies
// environment variables of the same names once at load (boostmvsnerfs_amd/_lib.py), in the open.
#include <atomic>
#include <string.h>

#include "bmv_common.hpp"

namespace bmv {

namespace {
struct Knob {
  const char* name;
  const char* what;
  std::atomic<int> value{0};
  std::atomic<bool> set{false};
};
Knob g_knobs[] = {
    {"BMV_SWEEP_WIN_CAP", "windowed sweep: LDS window budget in 64-byte records (tests: 48 forces the global-gather fallback)"},
    {"BMV_SWEEP_WIN_NH", "windowed sweep, 32 channels: 2 = both channel halves in one workgroup"},
    {"BMV_SWEEP_WIN_FLAGS", "windowed sweep ablations: 1 no fill, 2 no blend, 4 no store, 64 stamps"},
    {"BMV_RENDER_GRID", "fused renderer: workgroups (default 256 x waves per SIMD)"},
    {"BMV_RENDER_PC", "0 = fused renderer instead of the producer / consumer one"},
    {"BMV_RENDER_SPLIT", "0 = every chain of the fused MLP on fp32 MFMAs; default 1: its two-tile chains on the bf16 matrix pipe with three-piece fp32 operands at fp32 accuracy (the producer / consumer renderer and the stand-alone feat_ch 8 MLP)"},
    {"BMV_MVS_SPLIT", "0 = every layer of MVSNeRF's 6 x 128 MLP on fp32 MFMAs; default 1: 15 of its 17 weight chunks (every layer but the 20-input pts_bias) as bf16 MFMAs on three-piece fp32 operands at fp32 accuracy"},
    {"BMV_RENDER_PC_GRID", "producer / consumer renderer: workgroups (default 256: one per CU, all resident)"},
    {"BMV_MVS_SWEEP_AUX", "MVS padded sweep: cache-policy bits of its stores (default 0x102)"},
    {"BMV_SWEEP_QUAD_DBL", "1 = a workgroup of the quad plane sweep whose windows fit its share of the CU's LDS twice requests the next channel quad's windows before it blends the current one (two window sets, hand-written LDS reads); default 0: one set (measured round 6: the headline frame's windows do not fit twice at its occupancy, profiles/r6/sweep_double_buffer.txt)"},
    {"BMV_CONV2D_S_ROWS", "bf16 x 3 FeatureNet encoder convolutions (csrc/conv2d_s.hip): output rows a wave walks (4, 8; default 0: 8 when that gives >= 1536 waves)"},
    {"BMV_CONV0_S_ROWS", "bf16 x 3 first FeatureNet block (csrc/fpn_s.hip conv0_s_kernel): output rows a wave walks (9, 10, 12; default 0: the cheapest)"},
    {"BMV_FPN_S_ROWS", "bf16 x 3 fused top-down + smooth0 (csrc/fpn_s.hip): output rows a wave walks (6, 8, 9, 10, 12; default 0: the cheapest by waves per SIMD x rows)"},
    {"BMV_CONV_C4S_TZ", "bf16 x 3 first layers / heads (csrc/conv_c4s.hip): output planes a workgroup walks (2 or 4; default 0: the largest tile that leaves >= 1024 workgroups)"},
    {"BMV_CONV_C4S_RW", "... output rows per wave (4 or 2: tiles of 16 x 16 or 16 x 8; default 0: as above)"},
    {"BMV_CONV_SPLIT_TZ", "split-bf16 convolution: tile depth"},
    {"BMV_CONV_SPLIT_TX", "split-bf16 convolution: tile width (16 / 32)"},
    {"BMV_CONV_SPLIT_RW", "split-bf16 convolution: rows per wave"},
    {"BMV_CONV_SPLITK", "0 = no split-K for the small interior layers of the regularisers"},
    {"BMV_CONV_SPLITK_ROWS", "split-K layers: output rows per workgroup (4, 2, 1; default: the largest that gives >= 1024 workgroups)"},
    {"BMV_CONV_PAIR_ROWS", "row-paired 3-D layers: rows per wave (4: half-height tiles, 5: only below 1024 workgroups)"},
    {"BMV_CONV0_R", "fused first FeatureNet block: rows per wave (default 4)"},
    {"BMV_FPN_SMOOTH_R", "fused FPN + smooth0: rows per tile (default 8)"},
    {"BMV_FPN_SMOOTH_PERSIST", "fused FPN + smooth0: persistent workgroups per CU (0 = one workgroup per tile)"},
    {"BMV_SWEEP_QUAD_L0", "quad-planar sweep: default tuning variant where the source maps have twice the volume's resolution"},
    {"BMV_SWEEP_QUAD_L1", "quad-planar sweep: default tuning variant where the source maps have the volume's resolution"},
    {"BMV_DETERMINISTIC", "1 = the host takes the *_fixed scatter entry points (order-independent fixed-point accumulation): bit-reproducible gradients"},
    {"BMV_FPN_TOPDOWN_SPLIT", "FPN top-down step: workgroups sharing the output channels of a pixel (1, 2, 4)"},
};
constexpr int kNumKnobs = sizeof(g_knobs) / sizeof(g_knobs[0]);

Knob* find(const char* name) {
  if (!name) return nullptr;
  for (int i = 0; i < kNumKnobs; ++i)
    if (strcmp(name, g_knobs[i].name) == 0) return &g_knobs[i];
  return nullptr;
}
}  // namespace

int tuning(const char* name, int dflt) {
  const Knob* k = find(name);
  return (k && k->set.load(std::memory_order_acquire)) ? k->value.load(std::memory_order_relaxed) : dflt;
}

}  // namespace bmv

extern "C" {

int bmv_tuning_set(const char* name, int value) {
  auto* k = bmv::find(name);
  BMV_REQUIRE(k, "bmv_tuning_set: unknown switch '%s' (bmv_tuning_name lists them)", name ? name : "(null)");
  k->value.store(value, std::memory_order_relaxed);
  k->set.store(true, std::memory_order_release);
  return BMV_OK;
}

int bmv_tuning_clear(const char* name) {
  auto* k = bmv::find(name);
  BMV_REQUIRE(k, "bmv_tuning_clear: unknown switch '%s'", name ? name : "(null)");
  k->set.store(false, std::memory_order_release);
  return BMV_OK;
}

int bmv_tuning_get(const char* name, int* value, int* is_set) {
  auto* k = bmv::find(name);
  BMV_REQUIRE(k && value && is_set, "bmv_tuning_get: unknown switch '%s' or null pointer", name ? name : "(null)");
  *is_set = k->set.load(std::memory_order_acquire) ? 1 : 0;
  *value = k->value.load(std::memory_order_relaxed);
  return BMV_OK;
}

const char* bmv_tuning_name(int i) { return (i >= 0 && i < bmv::kNumKnobs) ? bmv::g_knobs[i].name : nullptr; }
const char* bmv_tuning_doc(int i) { return (i >= 0 && i < bmv::kNumKnobs) ? bmv::g_knobs[i].what : nullptr; }

}  // extern "C"
