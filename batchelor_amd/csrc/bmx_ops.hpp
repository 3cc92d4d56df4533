// Device-level building blocks of the merge loop (implemented in pairs.hip / correct.hip / legacy.hip).
#pragma once
#include "bmx_common.hpp"

namespace bmx {
}  // namespace bmx
