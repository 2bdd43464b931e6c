# Drop-in for koide3/fast_gicp's CMakeLists.txt: REPLACES its `### CUDA ###` block (CMakeLists.txt:113-141) when the
# device half is this repository's engine instead of the eleven .cu files. What the maintainer changes:
#   1. option(BUILD_VGICP_CUDA ...) stays (the C++ side keeps testing USE_VGICP_CUDA); add the two cache variables below.
#   2. `find_package(CUDA REQUIRED)` (CMakeLists.txt:36-40) is no longer needed: the block below only needs hipcc.
#   3. src/fast_gicp/cuda/*.cu are not compiled; FastVGICPCudaCore / NDTCudaCore come from integration/fast_vgicp_cuda_hip.cpp
#      (the two classes re-implemented on the C ABI), compiled by the HOST compiler.
# Nothing else in the tree changes: fast_vgicp_cuda.cpp / ndt_cuda.cpp still instantiate the templates that call the cores.
set(FAST_GICP_AMD_DIR "" CACHE PATH "checkout of this repository (fast_gicp_amd/csrc, include/fast_vgicp_hip.h, integration/)")
set(FVH_GPU_ARCH "gfx950" CACHE STRING "offload architecture")

if(BUILD_VGICP_CUDA)
  add_definitions(-DUSE_VGICP_CUDA)
  find_program(HIPCC hipcc HINTS /opt/rocm/bin REQUIRED)

  # the engine: one hipcc command (fvh_capi.hip includes every kernel header)
  set(FVH_LIB ${CMAKE_LIBRARY_OUTPUT_DIRECTORY})
  if(NOT FVH_LIB)
    set(FVH_LIB ${CMAKE_CURRENT_BINARY_DIR})
  endif()
  set(FVH_LIB ${FVH_LIB}/libfast_vgicp_hip.so)
  file(GLOB FVH_KERNEL_HEADERS ${FAST_GICP_AMD_DIR}/fast_gicp_amd/csrc/*.hpp)
  add_custom_command(
    OUTPUT ${FVH_LIB}
    COMMAND ${HIPCC} --offload-arch=${FVH_GPU_ARCH} -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -mllvm -disable-machine-licm
            -o ${FVH_LIB} ${FAST_GICP_AMD_DIR}/fast_gicp_amd/csrc/fvh_capi.hip -ldl
    DEPENDS ${FAST_GICP_AMD_DIR}/fast_gicp_amd/csrc/fvh_capi.hip ${FVH_KERNEL_HEADERS} ${FAST_GICP_AMD_DIR}/include/fast_vgicp_hip.h
    VERBATIM)
  add_custom_target(fast_vgicp_hip_build DEPENDS ${FVH_LIB})

  # the library the rest of the tree already links: same name, same role (cuda_add_library(fast_vgicp_cuda ...) before)
  add_library(fast_vgicp_cuda SHARED ${FAST_GICP_AMD_DIR}/integration/fast_vgicp_cuda_hip.cpp)
  target_include_directories(fast_vgicp_cuda PRIVATE include thirdparty/Eigen ${FAST_GICP_AMD_DIR}/include ${catkin_INCLUDE_DIRS})
  target_link_libraries(fast_vgicp_cuda ${FVH_LIB} ${catkin_LIBRARIES})
  add_dependencies(fast_vgicp_cuda fast_vgicp_hip_build)

  # add vgicp_cuda to libfast_gicp (unchanged from the reference)
  target_sources(fast_gicp PRIVATE
    src/fast_gicp/gicp/fast_vgicp_cuda.cpp
    src/fast_gicp/ndt/ndt_cuda.cpp
  )
  target_link_libraries(fast_gicp fast_vgicp_cuda)
  add_dependencies(fast_gicp fast_vgicp_cuda)
  if(catkin_FOUND)
    install(TARGETS fast_vgicp_cuda LIBRARY DESTINATION ${CATKIN_PACKAGE_LIB_DESTINATION})
    install(FILES ${FVH_LIB} DESTINATION ${CATKIN_PACKAGE_LIB_DESTINATION})
  elseif(ament_cmake_FOUND)
    install(TARGETS fast_vgicp_cuda LIBRARY DESTINATION lib)
    install(FILES ${FVH_LIB} DESTINATION lib)
  endif()
endif()
