// Scene code written for the reference includes "globals.h"; the definitions live in common.h.
#pragma once
#include "common.h"
