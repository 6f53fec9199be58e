// Internal: the public C ABI plus shared launch helpers.
#pragma once
#include "../../include/pianobart_hip.h"
