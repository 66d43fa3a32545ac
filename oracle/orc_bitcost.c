/* CPU oracle (test infrastructure only; parity unpinned, see orc.h).
 * Lambda-scaled MV-difference cost table, restated from
 * source/encoder/bitcost.cpp:31-102.  The table is an INPUT of the GPU path
 * (the host builds it with float math and uploads it verbatim), so its exact
 * float rounding never enters a GPU-vs-oracle comparison. */
#include "orc.h"
#include <math.h>

/* x265_lambda_tab[X265_LOOKAHEAD_QP]: QP 12 at 8-bit, 24 at 10-bit, 36 at
 * 12-bit (source/common/common.h:217-221; tables source/common/constants.cpp:
 * 34-50, 74-90, 114-130) */
double orc_lookahead_lambda(void)
{
#if ORC_DEPTH == 12
    return 256.0;
#elif ORC_DEPTH == 10
    return 16.0;
#else
    return 1.0;
#endif
}

void orc_bitcost_table(double lambda, uint16_t* out)
{
    uint16_t* c = out + 2 * ORC_BC_MAX_MV;
    const float log2_2 = 2.0f / logf(2.0f);
    for (int i = 0; i <= 2 * ORC_BC_MAX_MV; i++)
    {
        float bits = i ? logf((float)(i + 1)) * log2_2 + 1.718f : 0.718f;
        double v = bits * lambda + 0.5f;
        if (v > (double)((1 << 15) - 1))
            v = (double)((1 << 15) - 1);
        c[i] = c[-i] = (uint16_t)v;
    }
}

/* The float bit sizes behind BitCost::bitcost(mv, mvp): source/encoder/bitcost.cpp:86-101 */
void orc_bitsizes(float* out)
{
    float* c = out + 2 * ORC_BC_MAX_MV;
    const float log2_2 = 2.0f / logf(2.0f);
    c[0] = 0.718f;
    for (int i = 1; i <= 2 * ORC_BC_MAX_MV; i++)
        c[i] = c[-i] = logf((float)(i + 1)) * log2_2 + 1.718f;
}
