/* placeholder, filled in below */
#include "zada_oracle.h"
