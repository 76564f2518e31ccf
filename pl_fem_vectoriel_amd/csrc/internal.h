// Internal definitions shared by the host and device halves of libplfem_hip.so.
#pragma once
#include "symbolic.h"

struct plfem_symbolic {
  plfem::Symbolic S;
};
