// Host-side construction of the kinematic model tables of ik_kinematics.h (plain C++, shared by
// libpinkhip.so and the CPU wave emulator): validates a pinkhip_model_desc, derives the per-dof
// and ancestor tables, lays everything out in one buffer and points a ModelDev at it.
#pragma once

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pinkhip.h"

namespace pinkhip {

struct ModelImage {
  std::vector<char> bytes;  // what goes to the device, 8-byte aligned sections
  size_t off_parent, off_jtype, off_idx_q, off_idx_v, off_placement, off_axis, off_frame_joint,
      off_frame_placement, off_dof_joint, off_dof_sub, off_anc, off_q_min, off_q_max, off_v_max, off_root_joint,
      off_root_placement, off_ancr;
  int nj, nq, nv, nf, root_nv;
  bool has_relative = false;  // some frame slot is relative (frame_root_joint != -2)
};

inline std::string build_model_image(const pinkhip_model_desc &d, ModelImage &im) {
  if (d.nj < 1 || d.nj > 256) return "nj must be in 1..256";
  if (d.nv < 1 || d.nv > PINKHIP_MAX_NV) return "nv must be in 1..PINKHIP_MAX_NV";
  if (d.nf < 0 || d.nf > 64) return "nf must be in 0..64";
  if (!d.parent || !d.jtype || !d.idx_q || !d.idx_v || !d.placement || !d.axis || !d.q_min || !d.q_max || !d.v_max)
    return "model arrays must not be NULL";
  if (d.nf > 0 && (!d.frame_joint || !d.frame_placement)) return "frame arrays must not be NULL";
  int nq = 0, nv = 0;
  std::vector<int32_t> dof_joint(d.nv, 0), dof_sub(d.nv, 0);
  for (int j = 0; j < d.nj; ++j) {
    if (d.parent[j] >= j || d.parent[j] < -1) return "joints must be in topological order (parent < child)";
    const int t = d.jtype[j];
    const int jq = (t == PINKHIP_JOINT_FREE_FLYER) ? 7 : 1, jv = (t == PINKHIP_JOINT_FREE_FLYER) ? 6 : 1;
    if (t < 0 || t > 2) return "unknown joint type";
    if (d.idx_q[j] != nq || d.idx_v[j] != nv) return "idx_q / idx_v must be the running offsets";
    if (nv + jv > d.nv) return "joint tangent dimensions exceed nv";
    for (int k = 0; k < jv; ++k) {
      dof_joint[nv + k] = j;
      dof_sub[nv + k] = k;
    }
    nq += jq;
    nv += jv;
  }
  if (nq != d.nq || nv != d.nv) return "nq / nv do not match the joints";
  if (d.root_nv < 0 || d.root_nv > d.nv) return "root_nv out of range";
  std::vector<unsigned char> anc((size_t)d.nf * d.nj, 0);
  for (int f = 0; f < d.nf; ++f) {
    int j = d.frame_joint[f];
    if (j < -1 || j >= d.nj) return "frame_joint out of range";
    while (j >= 0) {
      anc[(size_t)f * d.nj + j] = 1;
      j = d.parent[j];
    }
  }
  // relative frame slots: the ancestors of the root frame's joint; an ordinary slot has none
  im.has_relative = false;
  std::vector<unsigned char> ancr((size_t)d.nf * d.nj, 0);
  std::vector<int32_t> root_joint(d.nf, -2);
  std::vector<double> root_placement((size_t)12 * d.nf, 0.0);
  if (d.frame_root_joint) {
    if (!d.frame_root_placement) return "frame_root_placement must not be NULL when frame_root_joint is given";
    for (int f = 0; f < d.nf; ++f) {
      int j = d.frame_root_joint[f];
      if (j < -2 || j >= d.nj) return "frame_root_joint out of range";
      root_joint[f] = j;
      if (j == -2) continue;
      im.has_relative = true;
      if (f >= 16) return "relative frame slots must be among the first 16 frames";  // (one lane per such slot in every kernel)
      std::memcpy(root_placement.data() + 12 * f, d.frame_root_placement + 12 * f, 12 * sizeof(double));
      while (j >= 0) {
        ancr[(size_t)f * d.nj + j] = 1;
        j = d.parent[j];
      }
    }
  }
  im.nj = d.nj;
  im.nq = d.nq;
  im.nv = d.nv;
  im.nf = d.nf;
  im.root_nv = d.root_nv;
  im.bytes.clear();
  auto put = [&](const void *src, size_t n) {
    const size_t off = im.bytes.size();
    im.bytes.resize(off + ((n + 7) & ~size_t(7)));
    if (n) std::memcpy(im.bytes.data() + off, src, n);
    return off;
  };
  im.off_parent = put(d.parent, 4 * d.nj);
  im.off_jtype = put(d.jtype, 4 * d.nj);
  im.off_idx_q = put(d.idx_q, 4 * d.nj);
  im.off_idx_v = put(d.idx_v, 4 * d.nj);
  im.off_placement = put(d.placement, 8 * 12 * d.nj);
  im.off_axis = put(d.axis, 8 * 3 * d.nj);
  im.off_frame_joint = put(d.frame_joint, 4 * d.nf);
  im.off_frame_placement = put(d.frame_placement, 8 * 12 * d.nf);
  im.off_dof_joint = put(dof_joint.data(), 4 * d.nv);
  im.off_dof_sub = put(dof_sub.data(), 4 * d.nv);
  im.off_anc = put(anc.data(), anc.size());
  im.off_q_min = put(d.q_min, 8 * d.nq);
  im.off_q_max = put(d.q_max, 8 * d.nq);
  im.off_v_max = put(d.v_max, 8 * d.nv);
  im.off_root_joint = put(root_joint.data(), 4 * d.nf);
  im.off_root_placement = put(root_placement.data(), 8 * 12 * d.nf);
  im.off_ancr = put(ancr.data(), ancr.size());
  return std::string();
}

// ModelDev view of an image located at `base` (device or host address).
template <class ModelDevT>
inline ModelDevT model_view(const ModelImage &im, const char *base) {
  ModelDevT m{};
  m.nj = im.nj;
  m.nq = im.nq;
  m.nv = im.nv;
  m.nf = im.nf;
  m.root_nv = im.root_nv;
  m.parent = reinterpret_cast<const int *>(base + im.off_parent);
  m.jtype = reinterpret_cast<const int *>(base + im.off_jtype);
  m.idx_q = reinterpret_cast<const int *>(base + im.off_idx_q);
  m.idx_v = reinterpret_cast<const int *>(base + im.off_idx_v);
  m.placement = reinterpret_cast<const double *>(base + im.off_placement);
  m.axis = reinterpret_cast<const double *>(base + im.off_axis);
  m.frame_joint = reinterpret_cast<const int *>(base + im.off_frame_joint);
  m.frame_placement = reinterpret_cast<const double *>(base + im.off_frame_placement);
  m.dof_joint = reinterpret_cast<const int *>(base + im.off_dof_joint);
  m.dof_sub = reinterpret_cast<const int *>(base + im.off_dof_sub);
  m.anc = reinterpret_cast<const unsigned char *>(base + im.off_anc);
  m.q_min = reinterpret_cast<const double *>(base + im.off_q_min);
  m.q_max = reinterpret_cast<const double *>(base + im.off_q_max);
  m.v_max = reinterpret_cast<const double *>(base + im.off_v_max);
  m.frame_root_joint = reinterpret_cast<const int *>(base + im.off_root_joint);
  m.frame_root_placement = reinterpret_cast<const double *>(base + im.off_root_placement);
  m.ancr = reinterpret_cast<const unsigned char *>(base + im.off_ancr);
  return m;
}

}  // namespace pinkhip
