// Test scaffolding: the enum ordinals the C ABI relies on (reference include/fast_gicp/gicp/gicp_settings.hpp:7-11).
#pragma once
namespace fast_gicp {
enum class RegularizationMethod { NONE, MIN_EIG, NORMALIZED_MIN_EIG, PLANE, FROBENIUS };
enum class NeighborSearchMethod { DIRECT27, DIRECT7, DIRECT1, DIRECT_RADIUS };
enum class VoxelAccumulationMode { ADDITIVE, ADDITIVE_WEIGHTED, MULTIPLICATIVE };
}  // namespace fast_gicp
