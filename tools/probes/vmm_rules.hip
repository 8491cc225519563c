// Which (offset, size) combinations does hipMemMap accept inside one reserved range?  (ROCm 7.2, MI355X)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
int main()
{
	hipMemAllocationProp prop = {};
	prop.type = hipMemAllocationTypePinned;
	prop.location.type = hipMemLocationTypeDevice;
	prop.location.id = 0;
	size_t gmin = 0, grec = 0;
	hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
	hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
	std::printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
	const size_t M = 1 << 20;
	const std::vector<std::vector<size_t>> plans = {{512, 256}, {256, 512}, {256, 256, 256}, {2048, 2024}, {2024, 2048}, {1024, 768},
	                                                {2048, 1024, 512, 256}, {768}, {2024}, {2048, 2048, 1976}};
	for (const auto &plan : plans) {
		size_t total = 0;
		for (size_t s : plan)
			total += s * M;
		void *va = nullptr;
		hipError_t e = hipMemAddressReserve(&va, total, 0, nullptr, 0);
		std::printf("reserve %zu MiB: %s at %p:", total / M, hipGetErrorString(e), va);
		size_t at = 0;
		for (size_t s : plan) {
			hipMemGenericAllocationHandle_t h;
			e = hipMemCreate(&h, s * M, &prop, 0);
			hipError_t e2 = e == hipSuccess ? hipMemMap((char *)va + at, s * M, 0, h, 0) : e;
			hipMemAccessDesc acc = {};
			acc.location = prop.location;
			acc.flags = hipMemAccessFlagsProtReadWrite;
			hipError_t e3 = e2 == hipSuccess ? hipMemSetAccess((char *)va + at, s * M, &acc, 1) : e2;
			std::printf("  [%zu MiB at +%zu: create %s, map %s, access %s]", s, at / M, hipGetErrorString(e), hipGetErrorString(e2), hipGetErrorString(e3));
			if (e3 == hipSuccess)
				hipMemset((char *)va + at, 0, s * M);
			(void)hipGetLastError();
			at += s * M;
		}
		std::printf("  sync: %s\n", hipGetErrorString(hipDeviceSynchronize()));
	}
	return 0;
}
