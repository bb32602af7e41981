// Internal declarations shared by the host orchestration (gsls_symbolic.cpp, gsls_api.cpp) and the
// HIP side (gsls_factor.hip, gsls_solve.hip).  Not part of the public ABI (include/gsls.h).
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/gsls.h"

namespace gsls {

// ---------------------------------------------------------------------------------------------
// Symbolic factorization.  Everything here is 0-based; the 1-based view the reference's akeep
// exposes (src/ssids/akeep.f90:25-76) is produced on demand by gsls_get_symbolic().
// ---------------------------------------------------------------------------------------------
struct Symbolic {
  int n = 0;
  int realn = 0;                 // structural rank (columns with at least one entry)
  int nnodes = 0;
  std::vector<int> perm;         // perm[var]  = pivot position   (ssids `order`, 0-based)
  std::vector<int> invp;         // invp[pos]  = variable
  std::vector<int> sptr;         // nnodes+1   first pivot position of each supernode
  std::vector<int> sparent;      // nnodes     parent supernode, nnodes for a root
  std::vector<int64_t> rptr;     // nnodes+1   offsets into rlist
  std::vector<int> rlist;        // sorted pivot positions (rows) of every front
  std::vector<int64_t> nptr;     // nnodes+1   offsets into nlist (pairs)
  std::vector<int64_t> nlist;    // 2*nz: (source index in val, destination in the m*n front)
  int64_t num_factor = 0;
  int64_t num_flops = 0;
  int maxfront = 0;
  int maxrow = 0;                // largest front row count (factor-time maxfront, cpu_iface.f90:84)
  int maxdepth = 0;

  // ---- schedule derived from the assembly tree (ours; no counterpart in akeep) ---------------
  std::vector<int> cptr, clist;  // children of each node (CSR, ascending)
  std::vector<int> level;        // height above the leaves (leaf = 0)
  int nlevels = 0;
  std::vector<int> lvlptr, lvlnodes;   // nodes grouped by level
  // child contribution row j (j = ncol..nrow-1 of the child) -> row index in the parent's front
  std::vector<int64_t> cmapptr;  // nnodes+1 offsets (length nrow-ncol per node)
  std::vector<int> cmap;
  std::vector<int64_t> loff;     // nnodes+1: element offset of each front's L block (ld = ldl[node])
  std::vector<int> ldl;          // leading dimension of each front's L block
  std::vector<int64_t> coff;     // nnodes+1: element offset of each contribution block ((m-n)^2); coff[nnodes] = arena size
  // the contribution arena is REUSED: a block lives from its front's level to its parent's (where it is assembled),
  // then its space goes back to a first-fit free list.  czero*: what has to be zeroed before each level runs.
  std::vector<int> czptr;        // nlevels+1: ranges of level l are [czptr[l], czptr[l+1])
  std::vector<int64_t> czoff, czlen;

  // ---- multi-GPU: subtree ownership (empty = the whole tree on one device) ----------------------
  int nranks = 1;
  std::vector<int> owner;        // nnodes: rank that factors the node, -1 = top part (rank 0, after the exchange)
  std::vector<int> cutroots;     // roots of the distributed subtrees, in exchange-buffer order

  int nrow(int node) const { return int(rptr[node + 1] - rptr[node]); }
  int ncol(int node) const { return sptr[node + 1] - sptr[node]; }
};

// returns a GSLS_* flag (0 ok, GSLS_ERROR_ORDER, GSLS_WARNING_ANAL_SINGULAR, ...)
// force_var (n flags by variable, may be null): variables that must share a supernode with their parent column
int symbolic_analyse(int n, const int64_t* ptr, const int32_t* row, int32_t* order, int ordering,
                     int nemin, Symbolic& S, const uint8_t* force_var = nullptr);

// fill-reducing ordering (nested dissection on the graph of A); writes perm[var] = position (0-based)
void order_amd(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow, std::vector<int>& perm);
void order_nested_dissection(int n, const std::vector<int64_t>& aptr, const std::vector<int>& arow,
                             std::vector<int>& perm);

// Split the assembly tree for `nranks` devices: independent subtrees are dealt to the ranks by decreasing
// work, everything above them (the top part) stays with rank 0 (cf. find_subtree_partition,
// src/ssids/anal.f90:284-459).  Fills S.owner / S.cutroots.
void shard_tree(Symbolic& S, int nranks);
void shard_layout(Symbolic& S, int rank);    // per-rank factor / arena offsets (only what the rank owns)

// offsets of the contribution blocks: reuse = true packs them by lifetime (single device), false lays them out one
// after the other (multi-GPU: the cut roots' blocks must survive until the exchange)
// gsls_scaling.cpp: the reference's internal scalings (spral/scaling.f90); lower triangle by columns, 0-based
int hungarian_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, bool scale_if_singular,
                        double* scaling);
int auction_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, double* scaling);
// matching-based ordering + its scaling (SSIDS ordering = 2, spral/match_order.f90); order[i] = position, 1-based
int match_order_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, int ordering, int32_t* order,
                    double* scaling, int32_t* npairs);
int equilib_scale_sym(int n, const int64_t* ptr, const int32_t* row, const double* val, double* scaling);

void layout_contrib(Symbolic& S, bool reuse);
void layout_contrib_auto(Symbolic& S);       // reuse when the blocks would not fit side by side comfortably

inline int align_ld(int m) { return (m + 1) & ~1; }   // 16-byte aligned columns

}  // namespace gsls
