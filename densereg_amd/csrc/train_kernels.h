// train_kernels.h -- training-side kernels (BatchReNorm batch statistics, backward passes, loss, Adam).
#pragma once
#include "dr_platform.h"
#include "kernels_misc.h"

namespace dr {
}  // namespace dr
