/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Rectangular linear sum assignment, restating the algorithm of the third-party dependency
 * the reference calls at src/d_fine/matcher.py:116 -- scipy.optimize.linear_sum_assignment
 * (scipy pinned to 1.15.1 in the reference's requirements.txt:22; not vendored in
 * /root/reference). Published algorithm: D. F. Crouse, "On implementing 2D rectangular
 * assignment algorithms", IEEE T-AES 52(4), 2016 -- shortest augmenting paths with dual
 * variables, as implemented in scipy/optimize/rectangular_lsap. Tie-breaking details that
 * decide WHICH optimum is returned are kept (column scan order `remaining = nc-1..0`,
 * preference for an unassigned column among equal path costs), so results are checked
 * index-for-index against scipy in tests/test_lsap.py.
 *
 * cost: row-major double [nr][nc]. Outputs min(nr,nc) pairs (row[k], col[k]) sorted by row.
 * Returns 0, or -1 infeasible (cost contains NaN / no finite path), -2 out of memory.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int64_t augmenting_path(int64_t nc, const double* cost, const double* u, const double* v,
                               int64_t* path, const int64_t* row4col, double* spc, int64_t i,
                               char* SR, char* SC, int64_t* remaining, double* p_min) {
  double min_val = 0;
  int64_t num_remaining = nc, sink = -1;
  for (int64_t it = 0; it < nc; it++) remaining[it] = nc - it - 1;
  for (int64_t j = 0; j < nc; j++) spc[j] = INFINITY;
  while (sink == -1) {
    int64_t index = -1;
    double lowest = INFINITY;
    SR[i] = 1;
    for (int64_t it = 0; it < num_remaining; it++) {
      int64_t j = remaining[it];
      double r = min_val + cost[i * nc + j] - u[i] - v[j];
      if (r < spc[j]) {
        path[j] = i;
        spc[j] = r;
      }
      if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
        lowest = spc[j];
        index = it;
      }
    }
    min_val = lowest;
    if (min_val == INFINITY) return -1;
    int64_t j = remaining[index];
    if (row4col[j] == -1) sink = j; else i = row4col[j];
    SC[j] = 1;
    remaining[index] = remaining[--num_remaining];
  }
  *p_min = min_val;
  return sink;
}

typedef struct { int64_t key, idx; } kv_t;
static int kv_cmp(const void* a, const void* b) {
  int64_t x = ((const kv_t*)a)->key, y = ((const kv_t*)b)->key;
  return (x > y) - (x < y);
}

int dfine_oracle_lsap(const double* cost_in, int64_t nr, int64_t nc, int64_t* row_out, int64_t* col_out) {
  if (nr == 0 || nc == 0) return 0;
  const int transpose = nc < nr;
  double* cost = (double*)malloc(sizeof(double) * nr * nc);
  if (!cost) return -2;
  if (transpose) {
    for (int64_t i = 0; i < nr; i++) for (int64_t j = 0; j < nc; j++) cost[j * nr + i] = cost_in[i * nc + j];
    int64_t t = nr; nr = nc; nc = t;
  } else {
    memcpy(cost, cost_in, sizeof(double) * nr * nc);
  }
  for (int64_t k = 0; k < nr * nc; k++)
    if (isnan(cost[k]) || cost[k] == -INFINITY) { free(cost); return -1; }

  double* u = (double*)calloc(nr, sizeof(double));
  double* v = (double*)calloc(nc, sizeof(double));
  double* spc = (double*)malloc(sizeof(double) * nc);
  int64_t* path = (int64_t*)malloc(sizeof(int64_t) * nc);
  int64_t* col4row = (int64_t*)malloc(sizeof(int64_t) * nr);
  int64_t* row4col = (int64_t*)malloc(sizeof(int64_t) * nc);
  int64_t* remaining = (int64_t*)malloc(sizeof(int64_t) * nc);
  char* SR = (char*)malloc(nr);
  char* SC = (char*)malloc(nc);
  int rc = 0;
  if (!u || !v || !spc || !path || !col4row || !row4col || !remaining || !SR || !SC) { rc = -2; goto done; }
  for (int64_t j = 0; j < nc; j++) { path[j] = -1; row4col[j] = -1; }
  for (int64_t i = 0; i < nr; i++) col4row[i] = -1;

  for (int64_t cur = 0; cur < nr; cur++) {
    double min_val;
    memset(SR, 0, nr);
    memset(SC, 0, nc);
    int64_t sink = augmenting_path(nc, cost, u, v, path, row4col, spc, cur, SR, SC, remaining, &min_val);
    if (sink < 0) { rc = -1; goto done; }
    u[cur] += min_val;
    for (int64_t i = 0; i < nr; i++) if (SR[i] && i != cur) u[i] += min_val - spc[col4row[i]];
    for (int64_t j = 0; j < nc; j++) if (SC[j]) v[j] -= min_val - spc[j];
    int64_t j = sink;
    for (;;) {
      int64_t i = path[j];
      row4col[j] = i;
      int64_t t = col4row[i]; col4row[i] = j; j = t;
      if (i == cur) break;
    }
  }
  if (transpose) {  /* rows of the caller = columns here: emit sorted by the caller's row */
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * nr);
    if (!kv) { rc = -2; goto done; }
    for (int64_t i = 0; i < nr; i++) { kv[i].key = col4row[i]; kv[i].idx = i; }
    qsort(kv, nr, sizeof(kv_t), kv_cmp);
    for (int64_t i = 0; i < nr; i++) { row_out[i] = kv[i].key; col_out[i] = kv[i].idx; }
    free(kv);
  } else {
    for (int64_t i = 0; i < nr; i++) { row_out[i] = i; col_out[i] = col4row[i]; }
  }
done:
  free(cost); free(u); free(v); free(spc); free(path); free(col4row); free(row4col); free(remaining); free(SR); free(SC);
  return rc;
}
