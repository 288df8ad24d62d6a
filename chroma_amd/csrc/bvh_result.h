// bvh_result.h -- what chroma_bvh_build (host, bvh_build.cpp) and chroma_bvh_build_device (bvh_device.hip) hand back
// behind the same opaque handle: chroma_bvh_fetch / _data / _free serve both.
#pragma once
#include <stdint.h>
#include <vector>
#include "uninit_vector.h"

namespace chroma_host {
struct Node { uint32_t x, y, z, w; };
struct BvhResult {
    uninit_vector<Node> nodes;
    std::vector<uint64_t> layer_bounds;   // nlayers + 1 entries
};
}  // namespace chroma_host
