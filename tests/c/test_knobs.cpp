// csrc/knobs.hpp on the CPU: every environment variable the library reads is parsed in one place, once per context.  The device
// list of eth_kzg_das_context_new (ETH_KZG_AMD_DEVICES: what lets an unchanged host use all GPUs of a node) and the table budget are
// what a deployment sets by hand: well-formed values, sloppy ones and garbage.
#include "knobs.hpp"
#include <cstdio>
#include <string>
using namespace kzg;

static int bad = 0, checks = 0;
static void expect(bool c, const char* what) {
    checks++;
    if (!c) { bad++; printf("MISMATCH %s\n", what); }
}
static Knobs with(const char* name, const char* value) {
    if (value) setenv(name, value, 1);
    else unsetenv(name);
    Knobs k = Knobs::from_env();
    unsetenv(name);
    return k;
}
static bool list_is(const Knobs& k, std::initializer_list<int> want) { return k.devices == std::vector<int>(want); }

int main() {
    for (const char* v : {"ETH_KZG_AMD_DEVICES", "ETH_KZG_AMD_DEVICE", "ETH_KZG_AMD_TABLE_GB", "ETH_KZG_AMD_DEVICE_BATCH_MAX", "ETH_KZG_AMD_ARENA_SIGNED",
                          "ETH_KZG_AMD_FAULT", "ETH_KZG_AMD_GLV_WINDOW", "ETH_KZG_AMD_PROGRESSIVE"})
        unsetenv(v);
    {
        const Knobs k = Knobs::from_env();  // nothing set: the defaults
        expect(k.devices.empty() && k.device == 0 && k.table_budget_gb == 0 && k.progressive && k.arena_signed && k.fault.empty() &&
               k.device_batch_max == 0 && k.glv_window == 0, "defaults");
    }
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "0"), {0}), "one device");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "0,1,2,3"), {0, 1, 2, 3}), "a list");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "3,1"), {3, 1}), "list order is kept");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "0,0"), {0, 0}), "an ordinal may repeat");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "all"), {-1}) && list_is(with("ETH_KZG_AMD_DEVICES", "ALL"), {-1}), "all");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "0,1,"), {0, 1}), "trailing comma");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "2,x,3"), {2}), "stops at garbage");
    expect(with("ETH_KZG_AMD_DEVICES", "").devices.empty() && with("ETH_KZG_AMD_DEVICES", "gpu0").devices.empty(), "empty / garbage: no list");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "1,-2,3"), {1, 3}) || list_is(with("ETH_KZG_AMD_DEVICES", "1,-2,3"), {1}), "a negative ordinal is not taken");
    expect(list_is(with("ETH_KZG_AMD_DEVICES", "5000"), {}), "an ordinal out of range is not taken");
    {
        std::string many;
        for (int i = 0; i < 100; i++) many += (i ? "," : "") + std::to_string(i % 8);
        expect(with("ETH_KZG_AMD_DEVICES", many.c_str()).devices.size() == 64, "at most 64 entries");
    }
    expect(with("ETH_KZG_AMD_DEVICE", "5").device == 5 && with("ETH_KZG_AMD_DEVICE", "-1").device == 0 && with("ETH_KZG_AMD_DEVICE", "x").device == 0, "single device");
    expect(with("ETH_KZG_AMD_TABLE_GB", "max").table_budget_gb < 0 && with("ETH_KZG_AMD_TABLE_GB", "44").table_budget_gb == 44 &&
               with("ETH_KZG_AMD_TABLE_GB", "0").table_budget_gb == 0 && with("ETH_KZG_AMD_TABLE_GB", "-3").table_budget_gb == 0 &&
               with("ETH_KZG_AMD_TABLE_GB", "lots").table_budget_gb == 0, "table budget");
    expect(with("ETH_KZG_AMD_DEVICE_BATCH_MAX", "128").device_batch_max == 128 && with("ETH_KZG_AMD_DEVICE_BATCH_MAX", "3").device_batch_max == 0, "device batch bound (>= 64)");
    expect(!with("ETH_KZG_AMD_ARENA_SIGNED", "0").arena_signed && with("ETH_KZG_AMD_ARENA_SIGNED", "1").arena_signed, "arena format");
    expect(!with("ETH_KZG_AMD_PROGRESSIVE", "0").progressive, "blocking constructor");
    expect(with("ETH_KZG_AMD_GLV_WINDOW", "15").glv_window == 15 && with("ETH_KZG_AMD_GLV_WINDOW", "99").glv_window == 0, "table width");
    {
        const Knobs k = with("ETH_KZG_AMD_FAULT", "constructor");
        expect(k.fault == "constructor", "fault injection name (copied: the variable is gone by now)");
    }
    printf("knobs: %d checks, %d mismatches\n", checks, bad);
    return bad ? 1 : 0;
}
