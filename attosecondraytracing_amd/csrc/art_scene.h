// art_scene.h -- the scene table: element descriptors and bundle views of MANY chains laid out for one launch.
//
// OEPlacement with a list-valued argument returns 10-11 chains that differ only in poses
// (ART/ModuleProcessing.py:203-239) and ARTmain traces them one after the other (ARTmain.py:304-342); the
// misalignment loop lists of ART/ModuleOpticalChain.py:371-657 do the same.  Here such a list is ONE launch:
// blockIdx.y selects the chain, blockIdx.x the 256-ray tile.  Descriptors no longer fit kernel arguments
// (3.1 KB per chain against a 4 KB limit), so they live in device memory as an array of ChainArgs -- the same
// struct the single-chain kernel takes by value -- and are fetched with scalar loads where they are used.
// The table is built on the host (scene_pack, no GPU call), uploaded by the caller and is all a launch reads:
// re-packing the poses into the same device buffer and replaying a captured HIP graph re-traces a modified scene
// without any host-side launch work.
//
// Shared by the HIP library and by the CPU twin under oracle/ (test infrastructure).
#pragma once
#include <stdint.h>
#include <string.h>

#include "art_device.h"

namespace art {

constexpr int kChainMax = 8;  // elements per fused launch (kernel-argument budget of the by-value form)

struct ChainArgs {
  ArtElementDesc e[kChainMax];
  ArtBundleView out[kChainMax];  // out[k].alive == NULL: no history for element k
  ArtBundleView in;
  ArtChainReadout ro;            // fused detector read-out of the last bundle (flags bit 1)
  int32_t n_elems;
  int32_t flags;                 // bit 0: some element carries Zernike or gridded defects; bit 1: `ro` is set
};
constexpr int kFlagDefects = 1, kFlagReadout = 2;
constexpr int kFlagMask = 4;     // scene header only: some chain contains a mask (selects the body of the launch)
constexpr int kFlagSharedIn = 8; // scene header only: every chain reads the SAME input bundle (selects the grid shape)
constexpr int kFlagSums = 16;    // scene header only: the tails form the analysis' sums (ArtChainReadout.sums), not a read-out

inline bool readout_ok(const ArtChainReadout& r) {
  const int outs = (r.X != nullptr) + (r.Y != nullptr) + (r.opl != nullptr);
  if (r.sums && (outs != 0 || r.lite)) return false;      // the sums tail has no per-ray outputs and no lite form
  return r.scratch && r.out24 && (outs == 0 || outs == 3);
}

constexpr uint32_t kSceneMagic = 0x41525453u;  // "ARTS"
struct SceneHeader {                           // first 64 bytes of an image
  uint32_t magic;
  int32_t n_chains, n_elems, n_segments, flags;
  int32_t reserved[11];
};
static_assert(sizeof(SceneHeader) == 64, "scene header is 64 bytes");

inline int scene_segments(int n_elems) { return (n_elems + kChainMax - 1) / kChainMax; }
inline int64_t scene_bytes(int n_chains, int n_elems) {
  return (int64_t)sizeof(SceneHeader) + (int64_t)scene_segments(n_elems) * n_chains * (int64_t)sizeof(ChainArgs);
}
inline const ChainArgs* scene_table(const void* image) {
  return reinterpret_cast<const ChainArgs*>(static_cast<const char*>(image) + sizeof(SceneHeader));
}

inline bool scene_view_ok(const ArtBundleView& v) {
  return v.ox && v.oy && v.oz && v.dx && v.dy && v.dz && v.path && v.incidence && v.alive;
}

// Returns flags (>= 0) or a negative ART_ERR_* code with a message in `err`.
// elems[c * n_elems + k], outs[c * n_elems + k], ins[c]; segment s of chain c sits at table[s * n_chains + c].
inline int scene_pack(const ArtElementDesc* elems, int n_chains, int n_elems, const ArtBundleView* ins,
                      const ArtBundleView* outs, const ArtChainReadout* ros, void* image, const char** err) {
  static const char* none = "";
  *err = none;
  if (!elems || !ins || !outs || !image) { *err = "NULL argument"; return ART_ERR_BAD_ARG; }
  if (n_chains <= 0 || n_chains > 65535) { *err = "n_chains must be in 1..65535"; return ART_ERR_BAD_ARG; }
  if (n_elems <= 0) { *err = "empty chain"; return ART_ERR_BAD_ARG; }
  const int S = scene_segments(n_elems);
  SceneHeader h;
  memset(&h, 0, sizeof(h));
  h.magic = kSceneMagic; h.n_chains = n_chains; h.n_elems = n_elems; h.n_segments = S;
  ChainArgs* tab = reinterpret_cast<ChainArgs*>(static_cast<char*>(image) + sizeof(SceneHeader));
  for (int c = 0; c < n_chains; ++c) {
    if (!scene_view_ok(ins[c])) { *err = "input bundle view has a NULL array"; return ART_ERR_BAD_ARG; }
    for (int s = 0; s < S; ++s) {
      ChainArgs& a = tab[(int64_t)s * n_chains + c];
      memset(&a, 0, sizeof(a));
      const int k0 = s * kChainMax;
      const int m = (n_elems - k0 < kChainMax) ? n_elems - k0 : kChainMax;
      a.n_elems = m;
      a.in = (s == 0) ? ins[c] : outs[(int64_t)c * n_elems + k0 - 1];
      for (int k = 0; k < m; ++k) {
        const ArtElementDesc& e = elems[(int64_t)c * n_elems + k0 + k];
        if (e.kind < 0 || e.kind >= ART_NUM_KINDS) { *err = "unknown optic kind"; return ART_ERR_BAD_ARG; }
        if (e.support_kind < 0 || e.support_kind > ART_SUP_RECTRECTHOLE) { *err = "unknown support kind"; return ART_ERR_BAD_ARG; }
        if (e.n_defects < 0 || e.n_defects > ART_MAX_DEFECTS || e.n_grid < 0 || e.n_grid > ART_MAX_DEFECTS) {
          *err = "too many defects on one mirror"; return ART_ERR_UNSUPPORTED;
        }
        if ((e.n_defects > 0 || e.n_grid > 0) && e.kind == ART_MASK) { *err = "a mask cannot carry defects"; return ART_ERR_BAD_ARG; }
        if ((e.n_defects > 0 && !e.zern) || (e.n_grid > 0 && !e.grid)) { *err = "defect count > 0 but its table is NULL"; return ART_ERR_BAD_ARG; }
        if (e.flags & ART_FLAG_ZERN_RECURRENCE) { *err = "an element carries Zernike tables in the recurrence layout: trace it with art_trace_element"; return ART_ERR_UNSUPPORTED; }
        a.e[k] = e;
        prepare_element(a.e[k]);
        if (e.n_defects > 0 || e.n_grid > 0) { a.flags |= kFlagDefects; h.flags |= kFlagDefects; }
        if (e.kind == ART_MASK) h.flags |= kFlagMask;
        const ArtBundleView& o = outs[(int64_t)c * n_elems + k0 + k];
        if (o.alive != nullptr && !scene_view_ok(o)) { *err = "history view partially NULL"; return ART_ERR_BAD_ARG; }
        a.out[k] = o;
      }
      if (ros && s == S - 1) {   // the read-out rides on the chain's last segment
        if (!readout_ok(ros[c])) { *err = "read-out: scratch/out24 missing, X/Y/opl partially NULL, or outputs / lite with sums"; return ART_ERR_BAD_ARG; }
        if ((ros[c].sums != 0) != (ros[0].sums != 0)) { *err = "read-outs of one scene must all be read-outs or all sums"; return ART_ERR_BAD_ARG; }
        a.ro = ros[c];
        a.flags |= kFlagReadout;
        h.flags |= kFlagReadout;
        if (ros[c].sums) h.flags |= kFlagSums;
      }
      // the segment's last bundle is the next segment's input (or the chain's result): it must exist
      if (!scene_view_ok(a.out[m - 1])) {
        *err = (s == S - 1) ? "the last output view is mandatory" : "chains longer than 8 need a view every 8th element";
        return ART_ERR_BAD_ARG;
      }
    }
  }
  bool shared = n_chains > 1;
  for (int c = 1; c < n_chains && shared; ++c) shared = memcmp(&ins[c], &ins[0], sizeof(ArtBundleView)) == 0;
  if (shared) h.flags |= kFlagSharedIn;
  memcpy(image, &h, sizeof(h));
  return h.flags;
}

}  // namespace art
