/* placeholder, filled in below */
#include "ck_oracle.h"
