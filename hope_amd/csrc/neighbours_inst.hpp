// The neighbour-search kernels (A6, A8) are most of the library's compile time.  neighbours.hip instantiates them;
// hope_hip.hip, which launches them, only declares the instantiations (extern template), so the two files compile
// side by side and an edit elsewhere does not rebuild them.  A kernel missing from this list still works: it is
// then instantiated where it is launched, as before.
#pragma once
#include "neighbours.hpp"
#include "grid_index.hpp"

#define HOPE_NEIGHBOUR_KERNELS(X)                                                                      \
  X(k_normals_grid<2>)                                                                                 \
  X(k_normals_grid<kNbCap>)                                                                            \
  X(k_normals_direct<2>)                                                                               \
  X(k_normals_direct<kNbCap>)                                                                          \
  X(k_normals_tile<2, kSuper>)                                                                         \
  X(k_normals_tile<2, kTile>)                                                                          \
  X(k_normals_tile<kNbCap, kSuper>)                                                                    \
  X(k_normals_tile<kNbCap, kTile>)                                                                     \
  X(k_knn_direct<8>)                                                                                   \
  X(k_knn_direct<kMaxKnn>)                                                                             \
  X(k_knn_tile<8, kSuper>)                                                                             \
  X(k_knn_tile<8, kTile>)                                                                              \
  X(k_knn_tile<kMaxKnn, kSuper>)                                                                       \
  X(k_knn_tile<kMaxKnn, kTile>)                                                                        \
  X(k_knn_todo<CoarseCols, 8>)                                                                         \
  X(k_knn_todo<CoarseCols, kMaxKnn>)                                                                   \
  X(k_knn_todo_threads<8>)

#define HOPE_KERNEL_INSTANTIATE(...) template __global__ decltype(__VA_ARGS__) __VA_ARGS__;
#define HOPE_KERNEL_ELSEWHERE(...) extern template __global__ decltype(__VA_ARGS__) __VA_ARGS__;
