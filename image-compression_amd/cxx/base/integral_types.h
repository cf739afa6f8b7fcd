// base/integral_types.h -- fixed-width integer names used throughout the Compressor API
// (uint8, uint32, ... in the global namespace).  Drop-in for the reference's header of the same
// path: same typedef names and limit constants, expressed through <stdint.h>.
#ifndef BASE_INTEGRAL_TYPES_H_
#define BASE_INTEGRAL_TYPES_H_

#include <stdint.h>

typedef int8_t int8;
typedef int16_t int16;
typedef int32_t int32;
typedef long long int64;  // NOLINT: callers print these with %lld
typedef signed char schar;

typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef unsigned long long uint64;  // NOLINT

typedef int32 char32;
typedef unsigned long uword_t;  // NOLINT

#define GG_LONGLONG(x) x##LL
#define GG_ULONGLONG(x) x##ULL
#define GG_LL_FORMAT "ll"

static const uint8 kuint8max = UINT8_MAX;
static const uint16 kuint16max = UINT16_MAX;
static const uint32 kuint32max = UINT32_MAX;
static const uint64 kuint64max = UINT64_MAX;
static const int8 kint8min = INT8_MIN;
static const int8 kint8max = INT8_MAX;
static const int16 kint16min = INT16_MIN;
static const int16 kint16max = INT16_MAX;
static const int32 kint32min = INT32_MIN;
static const int32 kint32max = INT32_MAX;
static const int64 kint64min = INT64_MIN;
static const int64 kint64max = INT64_MAX;

#endif  // BASE_INTEGRAL_TYPES_H_
