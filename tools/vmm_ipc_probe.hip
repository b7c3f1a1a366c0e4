// Probe: can two PROCESSES share a HIP virtual-memory allocation (hipMemExportToShareableHandle as a POSIX
// file descriptor, passed over a Unix socket, hipMemImportFromShareableHandle in the peer)?  This is the
// mechanism a partitioned item table would need under one process per GPU.
// hipcc --offload-arch=gfx950 -O2 tools/vmm_ipc_probe.hip -o tools/bin/vmm_ipc_probe
#include <hip/hip_runtime.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("[%d] FAIL %s -> %s\n", getpid(), #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(float* p, size_t n, float base) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = base + (float)i; }
__global__ void addone(float* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] += 1.0f; }
static int send_fd(int sock, int fd) {
    char buf[1] = {0}; struct iovec io = {buf, 1}; char c[CMSG_SPACE(sizeof(int))]; memset(c, 0, sizeof(c));
    struct msghdr msg = {}; msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = c; msg.msg_controllen = sizeof(c);
    struct cmsghdr* cm = CMSG_FIRSTHDR(&msg); cm->cmsg_level = SOL_SOCKET; cm->cmsg_type = SCM_RIGHTS; cm->cmsg_len = CMSG_LEN(sizeof(int));
    memcpy(CMSG_DATA(cm), &fd, sizeof(int)); return sendmsg(sock, &msg, 0) < 0;
}
static int recv_fd(int sock) {
    char buf[1]; struct iovec io = {buf, 1}; char c[CMSG_SPACE(sizeof(int))];
    struct msghdr msg = {}; msg.msg_iov = &io; msg.msg_iovlen = 1; msg.msg_control = c; msg.msg_controllen = sizeof(c);
    if (recvmsg(sock, &msg, 0) < 0) return -1;
    int fd; memcpy(&fd, CMSG_DATA(CMSG_FIRSTHDR(&msg)), sizeof(int)); return fd;
}
int main() {
    int sv[2]; if (socketpair(AF_UNIX, SOCK_STREAM, 0, sv)) return 2;
    const size_t bytes = (size_t)(getenv("PROBE_BYTES") ? atol(getenv("PROBE_BYTES")) : (1 << 21)), n = bytes / 4;
    pid_t pid = fork();
    hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    prop.requestedHandleTypes = hipMemHandleTypePosixFileDescriptor;
    hipMemAccessDesc acc = {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (pid == 0) { /* child: importer */
        close(sv[0]);
        hipMemGenericAllocationHandle_t mine; CK(hipMemCreate(&mine, bytes, &prop, 0));
        void* q = nullptr; CK(hipMemAddressReserve(&q, 4 * bytes, 0, nullptr, 0));
        CK(hipMemMap((char*)q + bytes, bytes, 0, mine, 0));
        int myfd = -1; CK(hipMemExportToShareableHandle(&myfd, mine, hipMemHandleTypePosixFileDescriptor, 0));
        printf("child: own export fd %d\n", myfd);
        int fd = recv_fd(sv[1]); if (fd < 0) { printf("child: no fd\n"); return 3; }
        printf("child: received fd %d\n", fd);
        hipMemGenericAllocationHandle_t h;
        CK(hipMemImportFromShareableHandle(&h, (void*)(uintptr_t)fd, hipMemHandleTypePosixFileDescriptor));
        void* p = nullptr; CK(hipMemAddressReserve(&p, bytes, 0, nullptr, 0));
        CK(hipMemMap(p, bytes, 0, h, 0)); CK(hipMemSetAccess(p, bytes, &acc, 1));
        std::vector<float> host(n); CK(hipMemcpy(host.data(), p, bytes, hipMemcpyDeviceToHost));
        size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += host[i] != 7.0f + (float)i;
        printf("child: read %zu mismatches of %zu\n", bad, n);
        addone<<<(unsigned)((n + 255) / 256), 256>>>((float*)p, n); CK(hipDeviceSynchronize());
        char ok = bad == 0; write(sv[1], &ok, 1);
        return bad != 0;
    }
    close(sv[1]);
    hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, bytes, &prop, 0));
    void* p = nullptr; CK(hipMemAddressReserve(&p, bytes, 0, nullptr, 0));
    CK(hipMemMap(p, bytes, 0, h, 0)); CK(hipMemSetAccess(p, bytes, &acc, 1));
    fill<<<(unsigned)((n + 255) / 256), 256>>>((float*)p, n, 7.0f); CK(hipDeviceSynchronize());
    int fd = -1; CK(hipMemExportToShareableHandle(&fd, h, hipMemHandleTypePosixFileDescriptor, 0));
    if (send_fd(sv[0], fd)) { printf("parent: send failed\n"); return 4; }
    char ok = 0; read(sv[0], &ok, 1);
    int st = 0; waitpid(pid, &st, 0);
    std::vector<float> host(n); CK(hipMemcpy(host.data(), p, bytes, hipMemcpyDeviceToHost));
    size_t bad = 0; for (size_t i = 0; i < n; ++i) bad += host[i] != 8.0f + (float)i;
    printf("parent: child ok=%d, sees child's writes: %zu mismatches of %zu\n", (int)ok, bad, n);
    printf(bad == 0 && ok ? "VMM IPC OK\n" : "VMM IPC FAILED\n");
    return 0;
}
