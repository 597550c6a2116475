// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" driver around the UNMODIFIED reference sources, compiled in place
// from /root/reference/src by oracle/Makefile into oracle/_ref/libfslic_ref.so.
// It exists so that (1) the C restatement in oracle/slic_oracle.c can be validated
// against the real reference stage by stage, (2) tests/golden/ fixtures can be
// generated, and (3) bench.py can time the real reference (SlicAvx2 + OpenMP) as
// cpu_baseline.kind == "reference".
//
// Stage access: the reference keeps quad_image / assignment / S protected
// (src/context.h:37-52).  We include every std header context.h needs FIRST and
// then re-define `protected`/`private` for the reference headers only, so the
// reference .cpp files themselves are compiled untouched.
#include <vector>
#include <chrono>
#include <cassert>
#include <cstring>
#include <string>
#include <memory>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <iostream>
#include <iomanip>
#include <sstream>
#include <map>
#include <limits>
#include <stdexcept>
#include <cstdlib>

#define protected public
#define private public
#include "context.h"
#include "lsc.h"
#include "cca.h"
#undef protected
#undef private
#include "parallel.h"
#include "fast-slic.h"

namespace {
// preemptive mode of the next iterate calls of this thread (cfast_slic.pyx:183-184); off by default
thread_local bool g_preemptive = false;
thread_local float g_preemptive_thres = 0.05f;
template <typename Ctx>
void configure(Ctx* c, float compactness, float min_size_factor, int subsample_stride,
               int convert_to_lab, int manhattan, int num_threads) {
    // mirrors cfast_slic.pyx:179-187
    c->num_threads = num_threads;
    c->compactness = compactness;
    c->min_size_factor = min_size_factor;
    c->subsample_stride_config = (int16_t)subsample_stride;
    c->convert_to_lab = convert_to_lab != 0;
    c->preemptive = g_preemptive;
    c->preemptive_thres = g_preemptive_thres;
    c->manhattan_spatial_dist = manhattan != 0;
    c->debug_mode = false;
}
thread_local std::string g_timing;
}

extern "C" {

void ref_set_preemptive(int on, float thres) { g_preemptive = on != 0; g_preemptive_thres = thres; }

// cfast_slic.pyx:359-369
int ref_supports_arch(const char* arch) {
    fslic::ContextBuilder b(arch);
    return b.is_supported_arch() ? 1 : 0;
}

// cfast_slic.pyx:124-147 (SlicModel.initialize)
int ref_initialize_clusters(const char* arch, int H, int W, int K, const uint8_t* rgb, Cluster* clusters) {
    fslic::ContextBuilder b(arch);
    if (!b.is_supported_arch()) return -1;
    fslic::Context* c = b.build(H, W, K, rgb, clusters);
    c->initialize_clusters();
    delete c;
    return 0;
}

// cfast_slic.pyx:150-260 (SlicModel.iterate, integer SLIC path) with optional stage dumps:
//   lab_out      : H*W*4 bytes (L,a,b,0) copied from quad_image            (may be NULL)
//   prelabels_out: H*W u16, labels after full_assign, before connectivity  (may be NULL)
//   labels_out   : H*W u16, final labels (0xFFFF possible)
int ref_slic_iterate(const char* arch, int H, int W, int K, const uint8_t* rgb, Cluster* clusters,
                     int max_iter, float compactness, float min_size_factor, int subsample_stride,
                     int convert_to_lab, int manhattan, int num_threads,
                     uint16_t* labels_out, uint8_t* lab_out, uint16_t* prelabels_out) {
    fslic::ContextBuilder b(arch);
    if (!b.is_supported_arch()) return -1;
    fslic::Context* c = b.build(H, W, K, rgb, clusters);
    configure(c, compactness, min_size_factor, subsample_stride, convert_to_lab, manhattan, num_threads);
    c->initialize_state();
    c->iterate(labels_out, max_iter);
    g_timing = c->get_timing_report();
    if (lab_out) {
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++)
                for (int ch = 0; ch < 4; ch++)
                    lab_out[(size_t)(i * W + j) * 4 + ch] = c->quad_image.get(i, 4 * j + ch);
    }
    if (prelabels_out) {
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++)
                prelabels_out[(size_t)i * W + j] = c->assignment.get(i, j);
    }
    delete c;
    return 0;
}

// LSC path: cfast_slic.pyx:207-216 -> src/lsc-builder.cpp
int ref_lsc_iterate(const char* arch, int H, int W, int K, const uint8_t* rgb, Cluster* clusters,
                    int max_iter, float compactness, float min_size_factor, int subsample_stride,
                    int convert_to_lab, int manhattan, int num_threads,
                    uint16_t* labels_out, uint16_t* prelabels_out) {
    fslic::ContextLSCBuilder b(arch);
    if (!b.is_supported_arch()) return -1;
    fslic::ContextLSC* c = b.build(H, W, K, rgb, clusters);
    configure(c, compactness, min_size_factor, subsample_stride, convert_to_lab, manhattan, num_threads);
    c->initialize_state();
    c->iterate(labels_out, max_iter);
    g_timing = c->get_timing_report();
    if (prelabels_out) {
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++)
                prelabels_out[(size_t)i * W + j] = c->assignment.get(i, j);
    }
    delete c;
    return 0;
}

// RealDist family: cfast_slic.pyx:198-235.  kind: 0 ContextRealDist ('standard'), 1 ContextRealDistL2 ('l2'),
// 2 ContextRealDistNoQ ('noq', float_color as given)
int ref_realdist_iterate(int kind, int H, int W, int K, const uint8_t* rgb, Cluster* clusters,
                         int max_iter, float compactness, float min_size_factor, int subsample_stride,
                         int convert_to_lab, int manhattan, int num_threads, int float_color,
                         uint16_t* labels_out, uint16_t* prelabels_out) {
    fslic::ContextRealDist* c = nullptr;
    if (kind == 0) c = new fslic::ContextRealDist(H, W, K, rgb, clusters);
    else if (kind == 1) c = new fslic::ContextRealDistL2(H, W, K, rgb, clusters);
    else if (kind == 2) { fslic::ContextRealDistNoQ* q = new fslic::ContextRealDistNoQ(H, W, K, rgb, clusters); q->float_color = float_color != 0; c = q; }
    else return -1;
    configure(c, compactness, min_size_factor, subsample_stride, convert_to_lab, manhattan, num_threads);
    c->initialize_state();
    c->iterate(labels_out, max_iter);
    g_timing = c->get_timing_report();
    if (prelabels_out) {
        for (int i = 0; i < H; i++)
            for (int j = 0; j < W; j++)
                prelabels_out[(size_t)i * W + j] = c->assignment.get(i, j);
    }
    delete c;
    return 0;
}

const char* ref_last_timing_report() { return g_timing.c_str(); }

// cfast_slic.pyx:371-396 (module-level enforce_connectivity): K = max label + 1 is computed by the caller.
int ref_enforce_connectivity(uint16_t* labels, int H, int W, int K, int min_threshold, int num_threads) {
    fsparallel::Scope scope(num_threads);
    cca::ConnectivityEnforcer ce(labels, H, W, K, min_threshold);
    ce.execute(labels);
    return 0;
}

// ---- superpixel-graph utilities on a finished label map (src/fast-slic.cpp) ----
// The reference returns a heap-allocated Connectivity; flattened here to num[K] + neighbors[K][stride].
static void flatten_connectivity(Connectivity* conn, int K, int stride, int* num_out, uint32_t* nb_out) {
    for (int k = 0; k < K; k++) {
        num_out[k] = conn->num_neighbors[k];
        for (int j = 0; j < stride; j++) nb_out[(size_t)k * stride + j] = j < conn->num_neighbors[k] ? conn->neighbors[k][j] : 0u;
    }
    fast_slic_free_connectivity(conn);
}
// fast_slic_get_connectivity (src/fast-slic.cpp:16-78); stride is its max_conn = 12
int ref_get_connectivity(int H, int W, int K, const uint16_t* labels, int* num_out, uint32_t* nb_out) {
    flatten_connectivity(fast_slic_get_connectivity(H, W, K, labels), K, 12, num_out, nb_out);
    return 0;
}
// fast_slic_knn_connectivity (src/fast-slic.cpp:80-130); stride = num_neighbors
int ref_knn_connectivity(int H, int W, int K, const Cluster* clusters, int num_neighbors, int num_threads, int* num_out, uint32_t* nb_out) {
    fsparallel::Scope scope(num_threads);
    flatten_connectivity(fast_slic_knn_connectivity(H, W, K, clusters, (size_t)num_neighbors), K, num_neighbors, num_out, nb_out);
    return 0;
}
// fast_slic_get_mask_density / fast_slic_cluster_density_to_mask (src/fast-slic.cpp:141-168)
int ref_get_mask_density(int H, int W, int K, const Cluster* clusters, const uint16_t* labels, const uint8_t* mask, uint8_t* densities) {
    fast_slic_get_mask_density(H, W, K, clusters, labels, mask, densities);
    return 0;
}
int ref_cluster_density_to_mask(int H, int W, int K, const Cluster* clusters, const uint16_t* labels, const uint8_t* densities, uint8_t* result) {
    fast_slic_cluster_density_to_mask(H, W, K, clusters, labels, densities, result);
    return 0;
}

// src/cielab.h:286-333 tables, for digest checks
int ref_lab_tables(int* gamma256, int* lab8193);

int ref_S(int H, int W, int K) {
    Cluster dummy[1];
    (void)dummy;
    return (int)(int16_t)sqrt(H * W / K);   // src/context.h:60
}

}  // extern "C"

#include "cielab.h"
extern "C" int ref_lab_tables(int* gamma256, int* lab8193) {
    for (int i = 0; i < 256; i++) gamma256[i] = fast_cielab_cvt.srgb_gamma_tbl[i];
    for (int i = 0; i <= 8192; i++) lab8193[i] = fast_cielab_cvt.lab_tbl[i];
    return 0;
}
