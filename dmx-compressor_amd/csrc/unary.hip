// csrc/unary.hip — the remaining per-element approximator-slot functions for gfx950.
//
// Function ids of the reference (src/dmx/compressor/__init__.py:108-139; functional/approximate.py): SILU, QUICK_GELU,
// EXP evaluate the EXACT torch function (the vsimd approximations live in a private package: parity unpinned), and
// the one in-repo approximation, `experimental.silu` (functional/functions.py:7-21), is reproduced bit for bit:
//     relu(x.to(float16)) * scale                                   -> float16
// Exact-function contracts, as torch evaluates them on the reference's CPU path:
//   SILU        torch.nn.functional.silu: x / (1 + exp(-x)) in fp32, rounded once to the output dtype;
//   EXP         torch.exp in fp32, rounded once;
//   QUICK_GELU  transformers' QuickGELUActivation, `input * torch.sigmoid(1.702 * input)`, which runs IN THE INPUT
//               DTYPE: three roundings for 16-bit tensors (t = fl(1.702 x), s = fl(sigmoid t), y = fl(x s)) -- kept.
// GELU (erf / tanh) stays in elementwise.hip (GeluOp); dmxq_unary forwards kinds 0 / 1 to it.
// All of them are one read + one write per element on the streaming skeleton of stream.hpp.
#include <math.h>

#include "stream.hpp"
#include "unary_ops.hpp"

namespace dmxq {

template <int KIND>
static int launch_unary(const void* in, void* out, int dti, int dto, int64_t n, float param, hipStream_t s) {
  // 16-bit in -> same 16-bit out (the module path) and fp32 -> fp32; the mixed pairs through the fp32 form
#define DMXQ_U(I_, O_, F_) \
  if (dti == I_ && dto == O_) return launch_stream<I_, O_, UnaryOp<KIND, I_, F_>>(in, out, n, UnaryOp<KIND, I_, F_>{param}, s);
  DMXQ_U(DMXQ_BF16, DMXQ_BF16, true)
  DMXQ_U(DMXQ_F16, DMXQ_F16, true)
  DMXQ_U(DMXQ_F32, DMXQ_F32, false)
  DMXQ_U(DMXQ_BF16, DMXQ_F32, false)
  DMXQ_U(DMXQ_F16, DMXQ_F32, false)
  DMXQ_U(DMXQ_F32, DMXQ_BF16, true)
  DMXQ_U(DMXQ_F32, DMXQ_F16, true)
#undef DMXQ_U
  return DMXQ_ERR_BAD_ARG;
}

}  // namespace dmxq

using namespace dmxq;

extern "C" int dmxq_gelu(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int tanh_form, void* stream);

extern "C" int dmxq_unary(const void* in, void* out, int dtype_in, int dtype_out, int64_t n, int kind, float param,
                          void* stream) {
  if (!valid_dtype(dtype_in) || !valid_dtype(dtype_out) || n < 0) return DMXQ_ERR_BAD_ARG;
  if (kind < DMXQ_UNARY_GELU || kind > DMXQ_UNARY_SILU_EXPERIMENTAL) return DMXQ_ERR_BAD_ARG;
  if (kind == DMXQ_UNARY_SILU_EXPERIMENTAL && dtype_out != DMXQ_F16) return DMXQ_ERR_BAD_ARG;  // the reference returns float16
  if (n == 0) return DMXQ_OK;
  if (!in || !out) return DMXQ_ERR_BAD_ARG;
  hipStream_t s = (hipStream_t)stream;
  switch (kind) {
    case DMXQ_UNARY_GELU: return dmxq_gelu(in, out, dtype_in, dtype_out, n, 0, stream);
    case DMXQ_UNARY_GELU_TANH: return dmxq_gelu(in, out, dtype_in, dtype_out, n, 1, stream);
    case DMXQ_UNARY_SILU: return launch_unary<DMXQ_UNARY_SILU>(in, out, dtype_in, dtype_out, n, param, s);
    case DMXQ_UNARY_QUICK_GELU: return launch_unary<DMXQ_UNARY_QUICK_GELU>(in, out, dtype_in, dtype_out, n, param, s);
    case DMXQ_UNARY_EXP: return launch_unary<DMXQ_UNARY_EXP>(in, out, dtype_in, dtype_out, n, param, s);
    default: {
      // one output dtype (float16): only the input dtype varies
      if (dtype_in == DMXQ_F16) return launch_stream<DMXQ_F16, DMXQ_F16, UnaryOp<DMXQ_UNARY_SILU_EXPERIMENTAL, DMXQ_F16, false>>(in, out, n, {param}, s);
      if (dtype_in == DMXQ_F32) return launch_stream<DMXQ_F32, DMXQ_F16, UnaryOp<DMXQ_UNARY_SILU_EXPERIMENTAL, DMXQ_F32, false>>(in, out, n, {param}, s);
      return launch_stream<DMXQ_BF16, DMXQ_F16, UnaryOp<DMXQ_UNARY_SILU_EXPERIMENTAL, DMXQ_BF16, false>>(in, out, n, {param}, s);
    }
  }
}
