// base/logging.h -- debug-check macros.  The reference compiles these away unless _DEBUG is set;
// this backend always compiles them away (argument errors are reported through return values).
#ifndef BASE_LOGGING_H_
#define BASE_LOGGING_H_

#define ICAMD_NOOP_CHECK(...) ((void)0)
#define DCHECK(c) ICAMD_NOOP_CHECK(c)
#define DCHECK_EQ(a, b) ICAMD_NOOP_CHECK(a, b)
#define DCHECK_NE(a, b) ICAMD_NOOP_CHECK(a, b)
#define DCHECK_LT(a, b) ICAMD_NOOP_CHECK(a, b)
#define DCHECK_LE(a, b) ICAMD_NOOP_CHECK(a, b)
#define DCHECK_GT(a, b) ICAMD_NOOP_CHECK(a, b)
#define DCHECK_GE(a, b) ICAMD_NOOP_CHECK(a, b)

#endif  // BASE_LOGGING_H_
