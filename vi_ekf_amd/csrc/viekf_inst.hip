// viekf_inst.hip -- explicit instantiations of the fused-step kernels of group VIEKF_INST_GROUP (viekf_instances.hpp).
#include <hip/hip_runtime.h>
#define VIEKF_INSTANCES_ONLY
#include "viekf_instances.hpp"

#define RES_DEF(RB, NW, NS) VIEKF_RES_FLAVOURS(, RB, NW, NS)
#define TILE_DEF(NT, NW) VIEKF_TILE_FLAVOURS(, NT, NW)
#if VIEKF_INST_GROUP == 0
VIEKF_RES_LIST_0(RES_DEF)
#elif VIEKF_INST_GROUP == 1
VIEKF_RES_LIST_1(RES_DEF)
#elif VIEKF_INST_GROUP == 2
VIEKF_RES_LIST_2(RES_DEF)
#elif VIEKF_INST_GROUP == 3
VIEKF_RES_LIST_3(RES_DEF)
#elif VIEKF_INST_GROUP == 4
VIEKF_RES_LIST_4(RES_DEF)
#elif VIEKF_INST_GROUP == 5
VIEKF_RES_LIST_5(RES_DEF)
#elif VIEKF_INST_GROUP == 6
VIEKF_RES_LIST_6(RES_DEF)
#elif VIEKF_INST_GROUP == 7
VIEKF_TILE_LIST_7(TILE_DEF)
#else
#error "VIEKF_INST_GROUP out of range"
#endif
