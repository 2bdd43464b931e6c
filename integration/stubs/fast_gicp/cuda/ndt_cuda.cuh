#pragma once
#include "../../core_decls_after_edit.hpp"
