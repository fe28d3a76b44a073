/* C++ hosts written against the reference's cpp/src/sdft/sdft.h: add -I<repo>/include/cpp and keep
   `#include <sdft/sdft.h>`; it resolves to the facade over the C-ABI. */
#pragma once
#include "../../sdft/sdft.hpp"
