"""Lab: build aki_amd/lib/abl/lib_timing.so - the attention core with s_memtime stamps around each phase of the tile loop
(barrier wait / K reads + bias / score MFMAs / softmax / PV MFMAs) and of the prologue, written into the lse buffer of
(batch 0, head 0).  Read them with tools/attn_timing.py.  The stamps are patched into a copy of the kernel source by exact
string anchors (asserted), so the shipped kernel carries no instrumentation; update the anchors when the loop changes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(os.path.join(ROOT, 'aki_amd/csrc/mma_attn_bf16.hip')).read()
def rep(a,b):
    global s
    assert a in s, a[:70]
    s=s.replace(a,b,1)
rep("template <int NW>\n__global__","""__device__ __forceinline__ unsigned stamp() {
  __builtin_amdgcn_sched_barrier(0);
  const unsigned long long t = __builtin_readcyclecounter();
  const unsigned r = __builtin_amdgcn_readfirstlane((unsigned)t);
  __builtin_amdgcn_sched_barrier(0);
  return r;
}
template <int NW>
__global__""")
rep("  int stage = 0;\n  for (int j = 0; j < jend; ++j) {","""  unsigned* const dbg = (unsigned*)p.lse + ((g * 4 + wave) * 16) * 8;
  const bool rec = bh == 0 && lane == 0;
  if (rec) { dbg[15 * 8 + 0] = t_entry; dbg[15 * 8 + 1] = stamp(); dbg[15 * 8 + 2] = wq0; dbg[15 * 8 + 3] = jend; dbg[15 * 8 + 5] = t_sync1; dbg[15 * 8 + 6] = t_q; }
  int stage = 0;
  for (int j = 0; j < jend; ++j) {
    const unsigned ta0 = stamp();""")
rep("  issue_tile(0, 0);\n  if (L > 64) issue_tile(1, 1);\n  int wq0, hi_col;","  const unsigned t_entry = stamp();\n  issue_tile(0, 0);\n  if (L > 64) issue_tile(1, 1);\n  int wq0, hi_col;")
rep("  load_q(wq0 + l31);\n  const int row = wq0 + l31;\n","  load_q(wq0 + l31);\n  const unsigned t_sync1 = stamp();\n  const int row = wq0 + l31;\n")
rep("  const int jend = (hi_col + 63) >> 6;","  const unsigned t_q = stamp();\n  const int jend = (hi_col + 63) >> 6;")
rep("    __builtin_amdgcn_s_barrier();\n    if (j + 2 < jend) issue_tile(","    __builtin_amdgcn_s_barrier();\n    const unsigned ta = stamp();\n    unsigned tb = ta, tc = ta, td = ta, te = ta;\n    if (j + 2 < jend) issue_tile(")
rep("      const bool lane_covers = ","      asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");\n      tb = stamp();\n      const bool lane_covers = ")
rep("      // The V^T fragments do not depend on the softmax","      { float tmp; asm volatile(\"v_add_f32 %0, %1, %2\" : \"=v\"(tmp) : \"v\"(s0[15]), \"v\"(s1[15])); asm volatile(\"s_nop 0\" :: \"v\"(tmp)); }\n      tc = stamp();\n      // The V^T fragments do not depend on the softmax")
rep("      // every transposed read has to be back before its registers are touched","      td = stamp();\n      // every transposed read has to be back before its registers are touched")
rep("""          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
        }
      }
    }
    vb_next = valid_word(j + 1);
    if (++stage == NSTAGE) stage = 0;
  }""","""          o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vv), pf, o[dt], 0, 0, 0);
        }
      }
      { float tmp; asm volatile("v_add_f32 %0, %1, %2" : "=v"(tmp) : "v"(o[0][15]), "v"(o[2][15])); asm volatile("s_nop 0" :: "v"(tmp)); }
      te = stamp();
    }
    if (rec && j < 15) { dbg[j * 8 + 0] = ta0; dbg[j * 8 + 1] = ta; dbg[j * 8 + 2] = tb; dbg[j * 8 + 3] = tc; dbg[j * 8 + 4] = td; dbg[j * 8 + 5] = te; dbg[j * 8 + 6] = (unsigned)full_tile; }
    vb_next = valid_word(j + 1);
    if (++stage == NSTAGE) stage = 0;
  }
  if (rec) dbg[15 * 8 + 4] = stamp();""")
rep("    if (!skip) {\n      const char* Kb","    int full_tile = 0;\n    if (!skip) {\n      const char* Kb")
rep("      if (full) {\n        // no bias:","      full_tile = full ? 1 : (rowwise ? 3 : 2);\n      if (full) {\n        // no bias:")
rep("  if (p.lse && h == 0 && row < L) p.lse","  if (rec) dbg[15 * 8 + 7] = stamp();\n  if (false) p.lse")
os.makedirs(os.path.join(ROOT, "aki_amd/lib/abl"), exist_ok=True)
src = os.path.join(ROOT, "aki_amd/lib/abl/attn_timing.hip")
open(src, "w").write(s)
flags = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "aki_amd/csrc"), "-ffp-contract=off"]
obj = src.replace(".hip", ".o")
subprocess.check_call(["/opt/rocm/bin/hipcc"] + flags + ["-c", src, "-o", obj])
others = [os.path.join(ROOT, "aki_amd/lib/obj", n + ".o") for n in ("api", "gemm_bf16", "attn_nc_bf16", "decode", "train_kernels", "attn_bwd_bf16", "fp8_quant", "simple_f32", "aux_kernels")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "aki_amd/lib/abl/lib_timing.so")] + others + [obj])
print(os.path.join(ROOT, "aki_amd/lib/abl/lib_timing.so"))
