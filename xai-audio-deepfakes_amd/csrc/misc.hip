#include "common.h"
int advh_init_rest() { return ADVH_OK; }
