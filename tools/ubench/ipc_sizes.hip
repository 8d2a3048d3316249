// Which device allocations can be exported with hipIpcGetMemHandle on this driver (dmabuf IPC, HSA_ENABLE_IPC_MODE_LEGACY=0), by
// allocation kind and size?  (round 4: the communicator's 64 MiB and 128 MiB windows failed with "invalid argument", 32 MiB worked.)
// Second half: fork a child that opens the parent's handle for the sizes that exported, writes a pattern, parent verifies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

struct shared_t { hipIpcMemHandle_t h[64]; size_t bytes[64]; int ok[64]; volatile int stage; volatile int child_ok[64]; };

__global__ void k_fill(unsigned* p, size_t n, unsigned v) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; for (; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v + (unsigned)i; }

int main()
{
  shared_t* sh = (shared_t*)mmap(nullptr, sizeof(shared_t), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  memset(sh, 0, sizeof(*sh));
  pid_t child = fork();
  size_t const sizes[] = {1u << 20, 16u << 20, 32u << 20, (32u << 20) + 4096, 48u << 20, 64u << 20, 128u << 20, 512u << 20, (size_t)2 << 30, (size_t)6 << 30};
  int const ns = sizeof(sizes) / sizeof(sizes[0]);
  if (child != 0) {  // exporter
    std::vector<void*> ptrs;
    int k = 0;
    for (int kind = 0; kind < 3; ++kind)
      for (int i = 0; i < ns; ++i, ++k) {
        void* p = nullptr;
        hipError_t e = kind == 0 ? hipMalloc(&p, sizes[i]) : hipExtMallocWithFlags(&p, sizes[i], kind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached);
        hipError_t g = hipErrorUnknown;
        if (e == hipSuccess) g = hipIpcGetMemHandle(&sh->h[k], p);
        (void)hipGetLastError();
        sh->bytes[k] = sizes[i];
        sh->ok[k]    = (e == hipSuccess && g == hipSuccess);
        printf("kind %d (%s) %8.1f MiB: alloc %s, getHandle %s  ptr %p\n", kind, kind == 0 ? "hipMalloc" : kind == 1 ? "finegrained" : "uncached", sizes[i] / 1048576.0,
               hipGetErrorName(e), e == hipSuccess ? hipGetErrorName(g) : "-", p);
        ptrs.push_back(p);
      }
    fflush(stdout);
    sh->stage = 1;
    while (sh->stage != 2) usleep(1000);
    k = 0;
    for (int kind = 0; kind < 3; ++kind)
      for (int i = 0; i < ns; ++i, ++k) {
        if (!sh->ok[k]) continue;
        unsigned v[2] = {0, 0};
        size_t const n = sh->bytes[k] / 4;
        hipMemcpy(&v[0], ptrs[k], 4, hipMemcpyDeviceToHost);
        hipMemcpy(&v[1], (unsigned*)ptrs[k] + (n - 1), 4, hipMemcpyDeviceToHost);
        printf("kind %d %8.1f MiB: child open %s, parent sees %s\n", kind, sh->bytes[k] / 1048576.0, sh->child_ok[k] ? "ok" : "FAILED",
               (v[0] == 77u && v[1] == 77u + (unsigned)(n - 1)) ? "the child's pattern" : "something else");
      }
    int st; waitpid(child, &st, 0);
    return 0;
  }
  while (sh->stage != 1) usleep(1000);
  for (int k = 0; k < 3 * ns; ++k) {
    if (!sh->ok[k]) continue;
    void* p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, sh->h[k], hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) {
      k_fill<<<256, 256>>>((unsigned*)p, sh->bytes[k] / 4, 77u);
      e = hipDeviceSynchronize();
      sh->child_ok[k] = (e == hipSuccess);
      hipIpcCloseMemHandle(p);
    } else { (void)hipGetLastError(); fprintf(stderr, "child: open %d failed: %s\n", k, hipGetErrorName(e)); }
  }
  sh->stage = 2;
  return 0;
}
