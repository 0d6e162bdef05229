#include "ck_oracle.h"
