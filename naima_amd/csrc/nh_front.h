// nh_front.h -- argument blocks shared by the weights / step-front kernels (nh_core.hip)
// and the fused half-step kernel (nh_halfstep.hip)
#pragma once
#include "nh_common.h"
#include "nh_pdist.h"

struct pw_grids {
  const double* e[NH_MAX_GRIDS];
  const double* xg[NH_MAX_GRIDS];
  const double* lne[NH_MAX_GRIDS];  // ln e_eV per node, or NULL
  const double* lx[NH_MAX_GRIDS];   // ln(xg[i+1]/xg[i]) per segment, or NULL
  double* w[NH_MAX_GRIDS];
  double* dlw[NH_MAX_GRIDS];
  double scale[NH_MAX_GRIDS];
  int nG[NH_MAX_GRIDS];
  int off[NH_MAX_GRIDS + 1];  // node offsets of the grids in one walker's flat index
  int n;
};


struct front_args {
  const double* coords; const double* logp; const double* blk;
  int* cursor; int* done;
  int ns, ndim, lo, nloc;
  double* qT; double* factors;
  nh_hist* hist;
  nh_pack pk[NH_MAX_PACK]; int npk;
  int kind; const double* params;
  pw_grids G;
  nh_moment mom[NH_MAX_MOMENT]; int nmom;
  int mom_off[NH_MAX_GRIDS];  // LDS offset (nodes) of a grid's w/dlw copy, or -1
  int mom_nodes;              // LDS nodes in total
};

