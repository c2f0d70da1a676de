// One translation unit of the parallel build (krotov_amd/build.py): compiled once per kernel family with
// -DKH_TU=<KH_TU_* of kh_common.h>.  It holds the explicit instantiations kh_instances.inc lists for that unit and the
// family's non-template kernels (KH_DEFINES); krotov_hip.hip (-DKH_TU=KH_TU_MAIN) launches them through their handles.
#include <hip/hip_runtime.h>

#include "kh_common.h"

#if KH_TU == KH_TU_ALL || KH_TU == KH_TU_MAIN
#error "kh_tu.hip is a family unit: compile it with -DKH_TU=<one of the family ids of kh_common.h>"
#endif

#if KH_TU == KH_TU_GENERIC
#include "kh_generic.h"
#elif KH_TU == KH_TU_MINI
#include "kh_mini.h"
#elif KH_TU == KH_TU_TILE
#include "kh_tile64.h"
#elif KH_TU == KH_TU_Q2
#include "kh_tile64q2.h"
#elif KH_TU == KH_TU_TILEX
#include "kh_tile64x.h"
#elif KH_TU == KH_TU_STREAM
#include "kh_tile64s.h"
#elif KH_TU == KH_TU_ENS
#include "kh_ens.h"
#elif KH_TU == KH_TU_TILEN
#include "kh_tilen.h"
#elif KH_TU == KH_TU_COOP_STORE || KH_TU == KH_TU_COOP_UPDATE_A || KH_TU == KH_TU_COOP_UPDATE_B
#include "kh_coop.h"
#elif KH_TU == KH_TU_ELL_STORE || KH_TU == KH_TU_ELL_UPDATE_A || KH_TU == KH_TU_ELL_UPDATE_B
#include "kh_ell.h"
#else
#error "unknown KH_TU"
#endif

#include "kh_instances.inc"
