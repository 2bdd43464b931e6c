// Test scaffolding: reference include/fast_gicp/ndt/ndt_settings.hpp:6.
#pragma once
namespace fast_gicp {
enum class NDTDistanceMode { P2D, D2D };
}  // namespace fast_gicp
