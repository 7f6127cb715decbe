// lfx_abi_guard.h — no C++ exception leaves an extern "C" entry point: host-side bookkeeping (std::vector, std::string) can throw
// std::bad_alloc, and unwinding through a C, Rust or Python frame is undefined.  The entry points that return a status are
// function-try-blocks closed by LFX_ABI_CATCH; the two that return a byte count (negative = -(LFX_E_*)) by LFX_ABI_CATCH_NEG; the
// constructors (`int *status` as their last parameter, the object or null as their value) by LFX_ABI_CATCH_NEW.
#pragma once
#include <new>

#include "../../include/lfx.h"

#define LFX_ABI_CATCH                                       \
    catch (const std::bad_alloc &) { return LFX_E_OOM; }    \
    catch (...) { return LFX_E_DEVICE; }
#define LFX_ABI_CATCH_NEG                                   \
    catch (const std::bad_alloc &) { return -LFX_E_OOM; }   \
    catch (...) { return -LFX_E_DEVICE; }
#define LFX_ABI_CATCH_NEW                                                        \
    catch (const std::bad_alloc &) { if (status) *status = LFX_E_OOM; return nullptr; }  \
    catch (...) { if (status) *status = LFX_E_DEVICE; return nullptr; }
