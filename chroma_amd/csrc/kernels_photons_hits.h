// kernels_photons_hits.h -- photon-array kernels of chroma/cuda/propagate.cu (duplicate, count, copy, select), hit extraction, k_finalize_hits.
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// initial queue of GPUPhotons.propagate (chroma/gpu/photon.py:206-216): slot 0 unused counter,
// then photon ids with the ncopies clones of a photon next to each other.
__global__ void k_init_queue(uint32_t *queue, uint64_t n, uint32_t ncopies, uint32_t true_n)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) queue[0] = (uint32_t)n + 1u;      // slot 0 = tail index, as after a step that queued all n
    if (j < n) {
        uint32_t copy = (uint32_t)(j % ncopies);
        uint32_t idx = (uint32_t)(j / ncopies);
        queue[1 + j] = idx + copy * true_n;
    }
}

__global__ void k_set_word(uint32_t *p, uint32_t v) { *p = v; }


// OR of (flags & mask) over all photons -> one word (abort warning, photon.py:254)
__global__ void k_flags_or(const uint32_t *flags, uint64_t n, uint32_t mask, uint32_t *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) acc |= flags[i] & mask;
    if (__ballot(acc != 0)) {
        for (int off = 32; off > 0; off >>= 1) acc |= __shfl_down(acc, off);
        if (lane_id() == 0 && acc) atomicOr(out, acc);
    }
}

__device__ inline void copy_photon(const PhotonView &src, size_t i, const PhotonView &dst, size_t o)
{
    store3(dst.pos, o, load3(src.pos, i));
    store3(dst.dir, o, load3(src.dir, i));
    store3(dst.pol, o, load3(src.pol, i));
    dst.wavelengths[o] = src.wavelengths[i];
    dst.t[o] = src.t[i];
    dst.flags[o] = src.flags[i];
    dst.last_hit_triangles[o] = src.last_hit_triangles[i];
    dst.weights[o] = src.weights[i];
    dst.evidx[o] = src.evidx[i];
    if (dst.rng_counters && src.rng_counters) dst.rng_counters[o] = src.rng_counters[i];
}

// photon_duplicate (chroma/cuda/propagate.cu:13-52)
__global__ void k_photon_duplicate(PhotonView pv, int first_photon, int nthreads, int copies, int stride)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nthreads) return;
    size_t photon_id = (size_t)first_photon + id;
    for (int i = 1; i <= copies; i++) copy_photon(pv, photon_id, pv, photon_id + (size_t)stride * i);
}

// count_photons (propagate.cu:54-79): grid-stride, one atomic per block
__global__ __launch_bounds__(256) void
k_count_photons(const uint32_t *flags, int first_photon, int nthreads, uint32_t target_flag, uint32_t *counter)
{
    __shared__ uint32_t s_total;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nthreads; id += (long long)gridDim.x * blockDim.x)
        mine += (flags[first_photon + id] & target_flag) != 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (lane_id() == 0 && mine) atomicAdd(&s_total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_total) atomicAdd(counter, s_total);
}

__device__ inline uint32_t wave_reserve(uint32_t *counter, bool pred, bool &any)
{
    unsigned long long mask = __ballot(pred);
    any = mask != 0ull;
    if (!any) return 0;
    unsigned lane = lane_id();
    unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// copy_photons (propagate.cu:81-114): one atomic per block of COPY_ITEMS * 256 photons, as k_copy_hits below
__global__ __launch_bounds__(256) void
k_copy_photons(PhotonView src, PhotonView dst, int first_photon, int nthreads, uint32_t target_flag, uint32_t *counter)
{
    __shared__ uint32_t s_wave[256 / WAVE + 1];
    const long long base = (long long)blockIdx.x * (16 * 256);
    const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
    uint32_t take = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const long long id = base + (long long)k * 256 + threadIdx.x;
        if (id < nthreads && (src.flags[first_photon + id] & target_flag)) { take |= 1u << k; mine++; }
    }
    uint32_t incl = mine;
    for (int off = 1; off < WAVE; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (unsigned w = 0; w < 256 / WAVE; w++) { uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
        s_wave[256 / WAVE] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t off = s_wave[256 / WAVE] + s_wave[wave] + incl - mine;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (take & (1u << k)) copy_photon(src, (size_t)first_photon + (size_t)(base + (long long)k * 256 + threadIdx.x), dst, off++);
}

// copy_photon_queue (propagate.cu:116-144)
__global__ void k_copy_photon_queue(PhotonView src, PhotonView dst, int first_photon, int nthreads, const uint32_t *queue)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nthreads) return;
    size_t offset = (size_t)first_photon + id;
    copy_photon(src, queue[offset], dst, offset);
}


// count_photon_hits (propagate.cu:147-174): grid-stride, one atomic per block
__global__ __launch_bounds__(256) void
k_count_hits(GeoView g, const uint32_t *flags, const int32_t *last_hit, int first_photon, int nphotons,
             uint32_t detection_state, uint32_t *counter)
{
    __shared__ uint32_t s_total;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nphotons; id += (long long)gridDim.x * blockDim.x)
        mine += hit_channel(g, flags[first_photon + id], last_hit[first_photon + id], detection_state) >= 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (lane_id() == 0 && mine) atomicAdd(&s_total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_total) atomicAdd(counter, s_total);
}

// copy_photon_hits (propagate.cu:176-214).  A block looks at COPY_ITEMS * 256 photons and reserves its
// output span with ONE atomic (the reference's one atomic per detected photon -- or one per wave -- on a
// single word costs 18 ms for 1e8 photons: a hot word serves ~88 atomics per microsecond).
#define COPY_ITEMS 16
__global__ __launch_bounds__(256) void
k_copy_hits(GeoView g, PhotonView src, PhotonView dst, int32_t *channels, int first_photon, int nphotons,
            uint32_t detection_state, uint32_t *counter)
{
    __shared__ uint32_t s_wave[256 / WAVE + 1];
    const long long base = (long long)blockIdx.x * (COPY_ITEMS * 256);
    const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
    int ch[COPY_ITEMS];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        const long long id = base + (long long)k * 256 + threadIdx.x;
        ch[k] = -1;
        if (id < nphotons) ch[k] = hit_channel(g, src.flags[first_photon + id], src.last_hit_triangles[first_photon + id], detection_state);
        mine += ch[k] >= 0;
    }
    // exclusive prefix of `mine` over the block: wave scan, then the waves' totals through LDS
    uint32_t incl = mine;
    for (int off = 1; off < WAVE; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (unsigned w = 0; w < 256 / WAVE; w++) { uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
        s_wave[256 / WAVE] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t off = s_wave[256 / WAVE] + s_wave[wave] + incl - mine;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        if (ch[k] >= 0) {
            copy_photon(src, (size_t)first_photon + (size_t)(base + (long long)k * 256 + threadIdx.x), dst, off);
            channels[off] = ch[k];
            off++;
        }
    }
}

// per-channel hit count + earliest time (float bits; non-negative times only, cuda/daq.cu:5-20)
__global__ void k_channel_hits(GeoView g, const uint32_t *flags, const int32_t *last_hit, const float *t, uint64_t n,
                               uint32_t detection_state, uint32_t *hit_count, uint32_t *earliest)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int ch = -1;
    uint32_t tb = 0xFFFFFFFFu;
    if (i < n) {
        ch = hit_channel(g, flags[i], last_hit[i], detection_state);
        if (ch >= 0 && earliest) tb = __float_as_uint(t[i]);
    }
    // The hits of a wave that fall on ONE channel are added with one atomic (a detector of few channels -- the stress
    // geometry has one -- otherwise serialises every hit on a hot word: 4.4 ms for 3.9e5 hits); with thousands of
    // channels the lanes of a wave hardly ever agree, and each adds its own.
    const unsigned long long hitters = __ballot(ch >= 0);
    if (!hitters) return;
    const int first = __builtin_amdgcn_readlane(ch, (int)__builtin_ctzll(hitters));
    if (__ballot(ch >= 0 && ch != first) == 0ull) {
        uint32_t m = tb;
        for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
        if (lane_id() == (unsigned)__builtin_ctzll(hitters)) {
            atomicAdd(&hit_count[first], (uint32_t)__popcll(hitters));
            if (earliest) atomicMin(&earliest[first], m);
        }
    } else if (ch >= 0) {
        atomicAdd(&hit_count[ch], 1u);
        if (earliest) atomicMin(&earliest[ch], tb);
    }
}

// ---- the end of a chroma_propagate_hits call: ONE pass over the photons ---------------------------------------------------
// What the reference does in four passes after propagate -- the abort-flag reduction (gpu/photon.py:254), count_photon_hits,
// copy_photon_hits (propagate.cu:147-214) and, for the detector's channel arrays, a DAQ-like reduction -- happens here while a
// photon's final state is in registers anyway: a photon that ended in k_physics during this call left a 64-byte record at
// final_rec[id] (stamped with the call's epoch), which is unpacked into the caller's ten arrays (coalesced: every array gets
// whole lines); any other photon (terminal before the call, finished by the tail kernel, or still alive at max_steps) is read
// from the arrays.  Detected photons that belong to a channel are counted, compacted into `dst` with their
// channel (one atomic per block of COPY_ITEMS * 256 photons, as k_copy_hits: the order of the blocks is the order of their atomics), and bump the per-channel count / earliest-time arrays.
// final_rec == NULL: everything comes from the arrays (the fused form of k_count_hits + k_copy_hits + k_channel_hits).
__global__ __launch_bounds__(256) void
k_finalize_hits(GeoView g, PhotonView pv, const float4 *final_rec, uint32_t epoch, uint64_t n, HitsOut h,
                uint32_t *words /* [0] number of hits, [2] OR of the NAN_ABORT bits */, uint32_t tail_mark)
{
    __shared__ uint32_t s_wave[256 / WAVE + 1];
    // (a wave's 64 records are 4 KB in a row: they come in as four coalesced kilobytes and reach their lanes through LDS --
    //  a lane reading its own record's four float4 touches 64 lines per instruction, as in k_load_working the other way)
    __shared__ float4 s_stage[256 / WAVE][WAVE * 4];
    const long long base = (long long)blockIdx.x * (COPY_ITEMS * 256);
    const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
    float4 *stg = s_stage[wave];
    int ch[COPY_ITEMS];
    uint32_t mine = 0, from_record = 0, aborts = 0;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        const long long id = base + (long long)k * 256 + threadIdx.x;
        ch[k] = -1;
        uint32_t tb = 0xFFFFFFFFu;
        if (final_rec) {
            const long long wave_first = base + (long long)k * 256 + (long long)wave * WAVE;       // (wave-uniform)
            const long long nrec = min((long long)WAVE, (long long)n - wave_first);                  // records of this wave: <= 0 none
            const float4 *src = final_rec + 4 * (size_t)max(wave_first, 0ll);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 4; r++) { const int q = r * WAVE + (int)lane; if (q < 4 * nrec) stg[q] = src[q]; }
            __builtin_amdgcn_wave_barrier();
        }
        if (id < (long long)n) {
            uint32_t flags; int lh = -1; float t = 0.f;
            bool have = false;
            float4 f3 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (final_rec) { f3 = stg[4 * lane + 3]; have = __float_as_uint(f3.w) == epoch; }
            // (a photon the tail kernel holds -- it may be finishing it right now, on the other stream -- is the tail kernel's
            //  to store, to count and to report: k_mark_tail stamped its record before this kernel started)
            if (final_rec && tail_mark != 0u && __float_as_uint(f3.w) == tail_mark) {
                flags = 0u;
            } else
            if (have) {
                const float4 f0 = stg[4 * lane], f1 = stg[4 * lane + 1], f2 = stg[4 * lane + 2];
                store3(pv.pos, (size_t)id, mk3(f0.x, f0.y, f0.z));
                store3(pv.dir, (size_t)id, mk3(f1.x, f1.y, f1.z));
                store3(pv.pol, (size_t)id, mk3(f2.x, f2.y, f2.z));
                pv.wavelengths[id] = f0.w;
                pv.t[id] = f1.w;
                pv.weights[id] = f2.w;
                flags = __float_as_uint(f3.x);
                pv.flags[id] = flags;
                pv.rng_counters[id] = __float_as_uint(f3.y);
                lh = __float_as_int(f3.z);
                pv.last_hit_triangles[id] = lh;
                t = f1.w;
                from_record |= 1u << k;
            } else {
                flags = pv.flags[id];
                if (h.want && (flags & h.detection_state)) { lh = pv.last_hit_triangles[id]; t = pv.t[id]; }
            }
            aborts |= flags & CHROMA_NAN_ABORT;
            if (h.want) {
                ch[k] = hit_channel(g, flags, lh, h.detection_state);
                if (ch[k] >= 0) { mine++; tb = __float_as_uint(t); }
            }
        }
        if (h.want && h.hit_count) {
            // (the hits of a wave that fall on ONE channel are added with one atomic: see k_channel_hits)
            const int c = ch[k];
            const unsigned long long hitters = __ballot(c >= 0);
            if (hitters) {
                const int first = __builtin_amdgcn_readlane(c, (int)__builtin_ctzll(hitters));
                if (__ballot(c >= 0 && c != first) == 0ull) {
                    uint32_t m = tb;
                    for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
                    if (lane == (unsigned)__builtin_ctzll(hitters)) {
                        atomicAdd(&h.hit_count[first], (uint32_t)__popcll(hitters));
                        if (h.earliest) atomicMin(&h.earliest[first], m);
                    }
                } else if (c >= 0) {
                    atomicAdd(&h.hit_count[c], 1u);
                    if (h.earliest) atomicMin(&h.earliest[c], tb);
                }
            }
        }
    }
    if (__ballot(aborts != 0u)) {
        for (int off = 32; off > 0; off >>= 1) aborts |= __shfl_down(aborts, off);
        if (lane == 0 && aborts) atomicOr(words + 2, aborts);
    }
    if (!h.want) return;
    // exclusive prefix of `mine` over the block: wave scan, then the waves' totals through LDS (as k_copy_hits)
    uint32_t incl = mine;
    for (int off = 1; off < WAVE; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (unsigned w = 0; w < 256 / WAVE; w++) { uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
        s_wave[256 / WAVE] = total ? atomicAdd(words, total) : 0u;
    }
    __syncthreads();
    if (!h.channels) return;
    uint32_t off = s_wave[256 / WAVE] + s_wave[wave] + incl - mine;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        if (ch[k] >= 0) {
            if (off < h.capacity) {
                const size_t id = (size_t)(base + (long long)k * 256 + threadIdx.x);
                if (from_record & (1u << k)) {
                    // (64 contiguous bytes instead of nine sparse reads of the arrays just written)
                    const float4 *f = final_rec + 4 * id;
                    const float4 f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
                    store3(h.dst.pos, off, mk3(f0.x, f0.y, f0.z));
                    store3(h.dst.dir, off, mk3(f1.x, f1.y, f1.z));
                    store3(h.dst.pol, off, mk3(f2.x, f2.y, f2.z));
                    h.dst.wavelengths[off] = f0.w;
                    h.dst.t[off] = f1.w;
                    h.dst.flags[off] = __float_as_uint(f3.x);
                    h.dst.last_hit_triangles[off] = __float_as_int(f3.z);
                    h.dst.weights[off] = f2.w;
                    h.dst.evidx[off] = pv.evidx[id];
                } else {
                    copy_photon(pv, id, h.dst, off);
                }
                h.channels[off] = ch[k];
            }
            off++;
        }
    }
}
