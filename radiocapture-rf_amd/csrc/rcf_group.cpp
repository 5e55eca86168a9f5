// rcf_group.cpp -- grouped launches over front-ends (placeholder while the split settles)
#include "rcf_plan.h"
