/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY) for the reference's `Hungarian` TF op.
 *
 * Plain-C restatement of hungarian.cc (reference checkout, file:line cited per function).
 * hungarian.cc itself cannot be compiled here (it needs TensorFlow 0.12 headers and
 * Eigen, neither present and neither vendored), so this restatement is pinned against
 * the reference's own known-answer tests hungarian_tf_tests.py:9-90 (tests/test_hungarian.py).
 *
 * Bit-exactness rests on: float state, the same add/sub/min/compare sequence, the same
 * float-vs-double comparison types (EPSILON and 1.0 are double literals in the reference,
 * hungarian.cc:18,292,302), ordered-set iteration (std::set<int> -> ascending index scan)
 * and being compiled with -ffp-contract=off.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link this.
 *
 * Return codes (the reference aborts the process instead, hungarian.cc:62,126,148,158,187,448):
 *   0  solved
 *   1  outer iteration cap reached: partial matching written (hungarian.cc:363-377)
 *  -2  BFS pop cap               (hungarian.cc:124-127)
 *  -3  parent-walk cap           (hungarian.cc:146-150,156-160)
 *  -4  max-flow augment cap      (hungarian.cc:184-188)
 *  -5  N_S/T equalisation cap    (hungarian.cc:446-450)
 *  -1  bad arguments / allocation failure
 */
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define ORA_EPSILON 1e-6
#define ORA_MAX_ITER 1000
#define ORA_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define ORA_ABS(x) (((x) > 0) ? (x) : -(x))

typedef struct {
  int n;           /* flow-network nodes = n_x + n_y + 2 */
  float *capacity; /* n*n */
  float *flow;     /* n*n */
  float *residual; /* n*n */
  int *queue;      /* BFS queue storage */
  int *parent;
  unsigned char *mark;
} ora_net;

/* hungarian.cc:107-177 Augment */
static int ora_augment(ora_net *g) {
  const int n = g->n, s = 0, t = n - 1;
  int head = 0, tail = 0, found = 0;
  g->queue[tail++] = s;
  memset(g->mark, 0, (size_t)n);
  for (int v = 0; v < n; ++v) g->parent[v] = -1;

  for (int i = 0; tail > head && i <= ORA_MAX_ITER; ++i) {
    if (i == ORA_MAX_ITER) return -2;
    int v = g->queue[head++];
    g->mark[v] = 1;
    if (v == t) {
      found = 1;
      break;
    }
    for (int u = 0; u < n; ++u) {
      if (!g->mark[u] && g->residual[v * n + u] > 0) {
        g->queue[tail++] = u;
        g->parent[u] = v;
      }
    }
  }

  if (found) {
    float b = g->capacity[0];
    for (int k = 1; k < n * n; ++k)
      if (g->capacity[k] > b) b = g->capacity[k]; /* capacity.maxCoeff() */
    int v = t;
    for (int i = 0; g->parent[v] != -1 && i <= ORA_MAX_ITER; ++i) {
      if (i == ORA_MAX_ITER) return -3;
      b = ORA_MIN(b, g->residual[g->parent[v] * n + v]);
      v = g->parent[v];
    }
    v = t;
    for (int i = 0; g->parent[v] != -1 && i <= ORA_MAX_ITER; ++i) {
      if (i == ORA_MAX_ITER) return -3;
      int p = g->parent[v];
      if (g->capacity[p * n + v] > 0) {
        g->flow[p * n + v] += b;
      } else {
        g->flow[v * n + p] -= b;
      }
      g->residual[p * n + v] -= b;
      g->residual[v * n + p] += b;
      v = p;
    }
  }
  return found;
}

/* hungarian.cc:179-217 MaxFlow + MaxBipartiteMatching: matching <- X->Y block of max flow */
static int ora_max_bipartite_matching(ora_net *g, const float *graph, int n_x, int n_y,
                                      float *matching) {
  const int n = g->n, s = 0, t = n_x + n_y + 1, x0 = 1, y0 = n_x + 1;
  memset(g->capacity, 0, sizeof(float) * (size_t)n * n);
  for (int x = 0; x < n_x; ++x)
    for (int y = 0; y < n_y; ++y) g->capacity[(x0 + x) * n + (y0 + y)] = graph[x * n_y + y];
  for (int x = 0; x < n_x; ++x) g->capacity[s * n + x0 + x] = 1.0f;
  for (int y = 0; y < n_y; ++y) g->capacity[(y0 + y) * n + t] = 1.0f;
  memset(g->flow, 0, sizeof(float) * (size_t)n * n);
  memcpy(g->residual, g->capacity, sizeof(float) * (size_t)n * n);
  for (int i = 0;; ++i) {
    int r = ora_augment(g);
    if (r < 0) return r;
    if (!(r && i <= ORA_MAX_ITER)) break;
    if (i == ORA_MAX_ITER) return -4;
  }
  for (int x = 0; x < n_x; ++x)
    for (int y = 0; y < n_y; ++y) matching[x * n_y + y] = g->flow[(x0 + x) * n + (y0 + y)];
  return 0;
}

/* hungarian.cc:219-248 */
static int ora_is_saturate(const float *m, int n_x, int n_y) {
  if (n_x >= n_y) {
    for (int j = 0; j < n_y; ++j) {
      float sum = 0;
      for (int i = 0; i < n_x; ++i) sum += m[i * n_y + j];
      if (sum == 0) return 0;
    }
    return 1;
  }
  for (int i = 0; i < n_x; ++i) {
    float sum = 0;
    for (int j = 0; j < n_y; ++j) sum += m[i * n_y + j];
    if (sum == 0) return 0;
  }
  return 1;
}

/* hungarian.cc:335-488 MinWeightedBipartiteCover (one [n_x, n_y] problem) */
static int ora_solve_one(const float *w, int n_x, int n_y, float *M, float *c_x, float *c_y) {
  const int n = n_x + n_y + 2;
  int rc = 0;
  ora_net g;
  g.n = n;
  size_t nn = (size_t)n * n;
  float *fbuf = (float *)malloc(sizeof(float) * (3 * nn + (size_t)n_x * n_y));
  int *ibuf = (int *)malloc(sizeof(int) * ((size_t)(ORA_MAX_ITER + 2) * n + n));
  unsigned char *bbuf = (unsigned char *)malloc((size_t)n + n_x + 2 * (size_t)n_y);
  if (!fbuf || !ibuf || !bbuf) {
    free(fbuf);
    free(ibuf);
    free(bbuf);
    return -1;
  }
  g.capacity = fbuf;
  g.flow = fbuf + nn;
  g.residual = fbuf + 2 * nn;
  float *equality = fbuf + 3 * nn;
  g.queue = ibuf;
  g.parent = ibuf + (size_t)(ORA_MAX_ITER + 2) * n;
  g.mark = bbuf;
  unsigned char *S = bbuf + n;        /* subset of X */
  unsigned char *T = S + n_x;         /* subset of Y */
  unsigned char *N_S = T + n_y;       /* subset of Y */
  memset(S, 0, (size_t)n_x);
  memset(T, 0, (size_t)n_y);
  int sizeT = 0;

  /* :339-353 */
  for (int x = 0; x < n_x; ++x) {
    float mx = w[x * n_y];
    for (int y = 1; y < n_y; ++y)
      if (w[x * n_y + y] > mx) mx = w[x * n_y + y];
    c_x[x] = mx;
  }
  for (int y = 0; y < n_y; ++y) c_y[y] = 0.0f;
  for (int k = 0; k < n_x * n_y; ++k) M[k] = 0.0f;

  int next_match = 1;
  for (int i = 0; i <= ORA_MAX_ITER; ++i) {
    if (i == ORA_MAX_ITER) {
      rc = 1; /* :363-377: log and return the unfinished matching */
      break;
    }
    /* :309-325 GetEqualityGraph */
    for (int x = 0; x < n_x; ++x)
      for (int y = 0; y < n_y; ++y) {
        float d = c_x[x] + c_y[y] - w[x * n_y + y];
        equality[x * n_y + y] =
            (ORA_ABS(d) <= ORA_EPSILON && (c_x[x] > 0 || c_y[y] > 0)) ? 1.0f : 0.0f;
      }
    if (next_match) {
      rc = ora_max_bipartite_matching(&g, equality, n_x, n_y, M);
      if (rc < 0) break;
      if (ora_is_saturate(M, n_x, n_y)) {
        rc = 0;
        break;
      }
      for (int u = 0; u < n_x; ++u) {
        int my = -1; /* :299-307 GetMatchedY */
        for (int v = 0; v < n_y; ++v)
          if (M[u * n_y + v] == 1.0) {
            my = v;
            break;
          }
        if (my == -1) {
          memset(S, 0, (size_t)n_x);
          S[u] = 1;
          memset(T, 0, (size_t)n_y);
          sizeT = 0;
          break;
        }
      }
    }

    /* :250-263 neighbours of S in the equality graph */
    memset(N_S, 0, (size_t)n_y);
    int sizeN = 0;
    for (int v = 0; v < n_x; ++v)
      if (S[v])
        for (int u = 0; u < n_y; ++u)
          if (equality[v * n_y + u] > 0 && !N_S[u]) {
            N_S[u] = 1;
            ++sizeN;
          }

    int equal = (sizeN == sizeT);
    if (equal)
      for (int y = 0; y < n_y; ++y)
        if (N_S[y] && !T[y]) {
          equal = 0;
          break;
        }

    if (equal) { /* :415-443 */
      float a = FLT_MAX;
      for (int x = 0; x < n_x; ++x)
        if (S[x])
          for (int y = 0; y < n_y; ++y)
            if (!T[y]) a = ORA_MIN(a, c_x[x] + c_y[y] - w[x * n_y + y]);
      if (a < ORA_EPSILON) {
        next_match = 1;
        continue;
      }
      for (int x = 0; x < n_x; ++x)
        if (S[x]) c_x[x] -= a;
      for (int y = 0; y < n_y; ++y)
        if (T[y]) c_y[y] += a;
    } else { /* :444-483 */
      for (int j = 0; sizeN > sizeT && j <= ORA_MAX_ITER; ++j) {
        if (j == ORA_MAX_ITER) {
          rc = -5;
          goto done;
        }
        int y = -1;
        for (int c = 0; c < n_y; ++c)
          if (N_S[c] && !T[c]) {
            y = c;
            break;
          }
        int z = -1; /* :289-297 GetMatchedX */
        for (int u = 0; u < n_x; ++u)
          if (M[u * n_y + y] == 1.0) {
            z = u;
            break;
          }
        if (z == -1) {
          next_match = 1;
          break;
        }
        next_match = 0;
        S[z] = 1;
        for (int v = 0; v < n_y; ++v)
          if (equality[z * n_y + v] > 0.0 && !N_S[v]) {
            N_S[v] = 1;
            ++sizeN;
          }
        if (!T[y]) {
          T[y] = 1;
          ++sizeT;
        }
      }
    }
  }
done:
  free(fbuf);
  free(ibuf);
  free(bbuf);
  return rc;
}

/* hungarian.cc:506-537 ComputeHungarianBatch (B >= 1) ; 2-D form = B == 1 (:490-504).
 * w [B,n_x,n_y] -> matching [B,n_x,n_y], cover_x [B,n_x], cover_y [B,n_y].
 * Returns the most severe per-example code (negative beats 1 beats 0). */
int ora_hungarian_f32(const float *w, int B, int n_x, int n_y, float *matching, float *cover_x,
                      float *cover_y) {
  if (!w || !matching || !cover_x || !cover_y || B < 0 || n_x <= 0 || n_y <= 0) return -1;
  int worst = 0;
  for (int b = 0; b < B; ++b) {
    int rc = ora_solve_one(w + (size_t)b * n_x * n_y, n_x, n_y, matching + (size_t)b * n_x * n_y,
                           cover_x + (size_t)b * n_x, cover_y + (size_t)b * n_y);
    if (rc < 0) {
      if (worst >= 0 || rc < worst) worst = rc;
    } else if (rc > worst && worst >= 0) {
      worst = rc;
    }
  }
  return worst;
}
