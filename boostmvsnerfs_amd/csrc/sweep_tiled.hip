// LDS-tiled plane sweep (a3+a4).  Placeholder until the tiled kernel lands: the
// dispatcher in sweep.hip falls back to the direct-gather kernel on UNSUPPORTED.
#include "bmv_common.hpp"

extern "C" int bmv_sweep_tiled_launch(const float*, const float*, const float*, int, int, int, int, int, int, int, int,
                                      float*, hipStream_t) {
  return BMV_ERR_UNSUPPORTED;
}
