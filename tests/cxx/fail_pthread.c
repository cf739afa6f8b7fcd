/* TEST INFRASTRUCTURE (tests/test_gpu_r06.py): an LD_PRELOAD interposer that makes pthread_create fail with EAGAIN on demand --
 * how std::thread's constructor gets to throw std::system_error inside icamd_compress_batch /
 * icamd_encode_batch_sharded_device on a box where RLIMIT_NPROC does not bind (the tests run as root).
 * icamd_test_fail_pthread_after(n): the next n creations succeed, every later one fails; n < 0 switches it off. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <pthread.h>
#include <stdatomic.h>

static atomic_int g_allowed = -1;
static atomic_int g_refused = 0;

void icamd_test_fail_pthread_after(int n) { atomic_store(&g_allowed, n); }
int icamd_test_pthread_refusals(void) { return atomic_load(&g_refused); }

int pthread_create(pthread_t *thread, const pthread_attr_t *attr, void *(*start)(void *), void *arg) {
  static int (*real)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *) = 0;
  if (!real) real = (int (*)(pthread_t *, const pthread_attr_t *, void *(*)(void *), void *))dlsym(RTLD_NEXT, "pthread_create");
  int left = atomic_load(&g_allowed);
  while (left >= 0) {
    if (left == 0) {
      atomic_fetch_add(&g_refused, 1);
      return EAGAIN;
    }
    if (atomic_compare_exchange_weak(&g_allowed, &left, left - 1)) break;
  }
  return real(thread, attr, start, arg);
}
