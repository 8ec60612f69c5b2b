// Host unit test of the launch planner (libear_amd/csrc/curves.h): for every call length, object count,
// kernel family and tuning knob, the plan plan_mix makes fits the bus buffer earhip_render_create
// allocates from bus_samples_bound().  No HIP call is made: runs without a GPU.
#include <cstdio>

#include "curves.h"

using namespace earhip;

int main() {
  long bad = 0, plans = 0;
  for (int num_cus : {256, 304, 64}) {
    for (int use_mfma : {0, 1, 2, 3, 4}) {
      for (int tpw : {1, 4, 8}) {
        for (int waves : {4, 8}) {
          for (int nrt : {4, 8}) {
            earhip_ctx ctx;
            ctx.num_cus = num_cus;
            ctx.use_mfma = use_mfma;
            ctx.tiles_per_wg = tpw;
            ctx.max_waves = waves;
            ctx.nrt = nrt;
            for (int M : {8, 64, 256, 1024, 4096}) {
              for (int ncols : {10, 24, 48}) {
                const ColumnPlan cp = ColumnPlan::make(ncols);
                for (int B : {64, 512, 4096}) {
                  for (int max_blocks : {1, 7, 64, 256, 511, 1024}) {
                    if ((long)max_blocks * B >= (1L << 24)) continue;
                    for (int max_gsplit : {1, 16, 32}) {
                      const size_t cap = bus_samples_bound(&ctx, (size_t)max_blocks * B, max_gsplit);
                      for (int nb = 1; nb <= max_blocks; nb += (nb < 70 ? 1 : 13)) {
                        const int ns = nb * B;
                        for (int aligned : {0, 256, 512}) {
                          for (bool strict : {false, true}) {
                            const MixLaunch ml = plan_mix(&ctx, cp, M, ns, strict, max_gsplit, aligned, 1.0, 1.0f);
                            const size_t need = (((size_t)ns + 3) & ~(size_t)3) * ml.gsplit;
                            plans++;
                            if (need > cap || ml.gsplit < 1 || ml.gsplit > max_gsplit) {
                              if (bad < 10)
                                printf("cus %d mfma %d tpw %d waves %d nrt %d M %d cols %d B %d max %d nb %d gs %d/%d "
                                       "aligned %d strict %d: need %zu > cap %zu\n",
                                       num_cus, use_mfma, tpw, waves, nrt, M, ncols, B, max_blocks, nb, ml.gsplit,
                                       max_gsplit, aligned, (int)strict, need, cap);
                              bad++;
                            }
                          }
                        }
                      }
                    }
                  }
                }
              }
            }
          }
        }
      }
    }
  }
  printf("%ld plans outside the bus bound of %ld\n", bad, plans);
  return bad != 0;
}
