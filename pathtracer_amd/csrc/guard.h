// guard.h — nothing unwinds across the C ABI (include/ptcore.h): every entry point whose body can allocate runs inside one
// of these, which turn std::bad_alloc into PT_ERR_NOMEM and anything else into PT_ERR_INVALID.
#ifndef PTC_GUARD_H
#define PTC_GUARD_H

#include <exception>
#include <new>

#include "../../include/ptcore.h"

template <class F>
static inline int ptc_noexcept_code(F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) { return PT_ERR_NOMEM; }
    catch (...) { return PT_ERR_INVALID; }
}

#ifdef PTC_DEVICE_TYPES_H
// the same with the text left in pt_last_error(ctx) (ctx may be NULL: pt_last_error(NULL) then); setting the text can itself
// run out of memory, in which case the code alone goes back
template <class F>
static inline int ptc_guard(pt_ctx* ctx, const char* what, F&& f) noexcept {
    try { return f(); }
    catch (const std::bad_alloc&) {
        try { return ptc_fail(ctx, PT_ERR_NOMEM, std::string(what) + ": out of host memory"); } catch (...) { return PT_ERR_NOMEM; }
    }
    catch (const std::exception& e) {
        try { return ptc_fail(ctx, PT_ERR_INVALID, std::string(what) + ": " + e.what()); } catch (...) { return PT_ERR_INVALID; }
    }
    catch (...) {
        try { return ptc_fail(ctx, PT_ERR_INVALID, std::string(what) + ": unexpected exception"); } catch (...) { return PT_ERR_INVALID; }
    }
}
#endif

#endif
