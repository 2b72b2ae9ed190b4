// dn_hop_glw.hip -- the wavefront-per-stream instantiations of dn_hop.hip's hop_kernel (n_fft 1024: deep pipes, the saturated regime, the front-only launch
// of a split hop) as a translation unit of their own, compiled with LLVM's iterative-ILP scheduling strategy (see the head of dn_hop.hip).  The stamped
// diagnostic build keeps them in dn_hop.hip: this file is empty there.
#ifndef DN_PROBE
#define DN_HOP_TU_GLW 1
#include "dn_hop.hip"
#endif
