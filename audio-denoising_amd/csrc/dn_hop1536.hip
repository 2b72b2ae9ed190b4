// dn_hop1536.hip -- the n_fft-1536 instantiations of dn_hop.hip's kernels (hop_kernel, frame_kernel) as a translation unit of their own, compiled
// with the default scheduling strategy (see the head of dn_hop.hip).  The stamped diagnostic build keeps them in dn_hop.hip: this file is empty there.
#ifndef DN_PROBE
#define DN_HOP_TU_1536 1
#include "dn_hop.hip"
#endif
