// spec_registry.hip -- the plan-specialised sub-step kernels linked into this library (csrc/gen/*.hip, written by
// deepqmc_amd/codegen) and the host-side packer of their weight tapes.
#include "spec_device.h"

#include <cstdlib>
#include <cstring>

namespace dqmc {

#define DQMC_SPEC_KERNEL(name) const SpecKernel* spec_kernel_##name();
#include "gen/spec_list.inc"
#undef DQMC_SPEC_KERNEL

typedef const SpecKernel* (*SpecGetter)();
static const SpecGetter kSpecKernels[] = {
#define DQMC_SPEC_KERNEL(name) &spec_kernel_##name,
#include "gen/spec_list.inc"
#undef DQMC_SPEC_KERNEL
    nullptr};

const SpecKernel* find_spec_kernel(uint64_t hash) {
  const char* want = getenv("DQMC_SPEC_VARIANT");      // development: several generated variants of one program
  const SpecKernel* first = nullptr;
  for (const SpecGetter* k = kSpecKernels; *k; ++k) {
    const SpecKernel* sk = (*k)();
    if (sk->hash != hash) continue;
    if (!first) first = sk;
    if (want && strcmp(want, sk->name) == 0) return sk;
  }
  return first;
}

// piece pl (0..2) of the three-bf16 split of a float (round to nearest even, exact residuals: common.h bf_split8)
static uint16_t bf16_piece(float v, int pl) {
  auto rne = [](float f) -> uint16_t {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  };
  auto up = [](uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
  uint16_t h = rne(v);
  for (int k = 0; k < pl; ++k) { v = v - up(h); h = rne(v); }
  return h;
}

// tape [tape_bytes / 4] words: the fragments in consumption order (SpecTapeEntry), zero padded to whole ring stages
void spec_pack_tape(const SpecKernel& k, const float* w, uint32_t* tape) {
  memset(tape, 0, (size_t)k.tape_bytes);
  size_t frag = 0;
  for (int e = 0; e < k.n_entries; ++e) {
    const SpecTapeEntry& t = k.entries[e];
    const int32_t* map = k.maps + t.map;
    if (t.kind == 0) {
      for (int pl = 0; pl < 3; ++pl) {
        uint32_t* f = tape + (frag + pl) * 256;
        for (int lane = 0; lane < 64; ++lane) {
          const int c = lane & 15, g = lane >> 4;
          for (int j = 0; j < 4; ++j) {
            uint32_t word = 0;
            for (int h = 0; h < 2; ++h) {
              const int row = map[8 * g + 2 * j + h];
              const float v = (row >= 0 && c < t.ncol) ? w[(size_t)t.w_off + (size_t)row * t.ldw + t.col0 + c] * t.scale : 0.0f;
              word |= (uint32_t)bf16_piece(v, pl) << (16 * h);
            }
            f[lane * 4 + j] = word;
          }
        }
      }
      frag += 3;
    } else {
      uint32_t* f = tape + frag * 256;
      for (int q = 0; q < 256; ++q) {
        const float v = map[q] >= 0 ? w[map[q]] * t.scale : 0.0f;
        memcpy(f + q, &v, 4);
      }
      frag += 1;
    }
  }
}

}  // namespace dqmc
