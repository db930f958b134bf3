// Test infrastructure: the slice of googletest's interface that the reference's suites use (tests/test_module_apply.cpp,
// tests/test_module_spmv_spmspv.cpp, tests/test_app.cpp: TEST, ASSERT_EQ, ASSERT_TRUE, testing::InitGoogleTest, RUN_ALL_TESTS),
// so that those files compile UNMODIFIED against include/graphlily and run on the GPU box, where googletest is not installed.
// Same observable behaviour for that slice: tests run in definition order, a failed ASSERT_* prints file:line and the expression,
// marks the test failed and returns from the CURRENT function (which is why the reference's verify() helpers are void), the
// summary lists failed tests and RUN_ALL_TESTS returns 1 if any failed.  --gtest_filter=PATTERN[:PATTERN...][-NEG...] with `*`
// wildcards is honoured (the runner skips the Synthesize / Clean fixtures' shell-outs that way when asked to).
#ifndef GRAPHLILY_HIP_MINI_GTEST_H_
#define GRAPHLILY_HIP_MINI_GTEST_H_

#include <chrono>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace testing {

struct TestInfo {
    const char *suite, *name;
    void (*fn)();
};
inline std::vector<TestInfo> &registry() {
    static std::vector<TestInfo> r;
    return r;
}
inline bool &current_failed() {
    static bool f = false;
    return f;
}
inline std::string &filter() {
    static std::string f = "*";
    return f;
}
struct Registrar {
    Registrar(const char *suite, const char *name, void (*fn)()) { registry().push_back(TestInfo{suite, name, fn}); }
};

inline bool glob(const char *p, const char *s) {
    if (!*p) return !*s;
    if (*p == '*') return glob(p + 1, s) || (*s && glob(p, s + 1));
    return *s && (*p == '?' || *p == *s) && glob(p + 1, s + 1);
}
inline bool any_of_patterns(const std::string &list, const std::string &full) {
    std::stringstream ss(list);
    std::string pat;
    while (std::getline(ss, pat, ':'))
        if (!pat.empty() && glob(pat.c_str(), full.c_str())) return true;
    return false;
}
inline bool selected(const std::string &full) {
    const std::string &f = filter();
    const size_t dash = f.find('-');
    const std::string pos = dash == std::string::npos ? f : f.substr(0, dash), neg = dash == std::string::npos ? "" : f.substr(dash + 1);
    return any_of_patterns(pos.empty() ? "*" : pos, full) && !(neg.size() && any_of_patterns(neg, full));
}

inline void InitGoogleTest(int *argc, char **argv) {
    for (int i = 1; i < *argc; i++)
        if (!strncmp(argv[i], "--gtest_filter=", 15)) filter() = argv[i] + 15;
}

inline int RunAll() {
    int ran = 0;
    std::vector<std::string> failed;
    for (const TestInfo &t : registry()) {
        const std::string full = std::string(t.suite) + "." + t.name;
        if (!selected(full)) continue;
        printf("[ RUN      ] %s\n", full.c_str());
        fflush(stdout);
        current_failed() = false;
        const auto t0 = std::chrono::steady_clock::now();
        t.fn();
        const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
        printf("%s %s (%ld ms)\n", current_failed() ? "[  FAILED  ]" : "[       OK ]", full.c_str(), ms);
        fflush(stdout);
        if (current_failed()) failed.push_back(full);
        ran++;
    }
    printf("[==========] %d tests ran.\n[  PASSED  ] %d tests.\n", ran, ran - (int)failed.size());
    for (const std::string &f : failed) printf("[  FAILED  ] %s\n", f.c_str());
    return failed.empty() ? 0 : 1;
}

}  // namespace testing

#define TEST(suite, name)                                                                          \
    static void gl_test_##suite##_##name();                                                        \
    static ::testing::Registrar gl_reg_##suite##_##name(#suite, #name, &gl_test_##suite##_##name); \
    static void gl_test_##suite##_##name()

#define GL_GTEST_FAIL_(text)                                              \
    do {                                                                  \
        std::cout << __FILE__ << ":" << __LINE__ << ": Failure\n" << text << std::endl; \
        ::testing::current_failed() = true;                               \
        return;                                                           \
    } while (0)

#define ASSERT_TRUE(cond)                                                      \
    do {                                                                       \
        if (!(cond)) GL_GTEST_FAIL_("Value of: " #cond "\n  Actual: false\nExpected: true"); \
    } while (0)

#define ASSERT_EQ(a, b)                                                        \
    do {                                                                       \
        if (!((a) == (b))) {                                                   \
            std::ostringstream gl_os;                                          \
            gl_os << "Expected equality of these values:\n  " #a "\n    Which is: " << (a) << "\n  " #b "\n    Which is: " << (b); \
            GL_GTEST_FAIL_(gl_os.str());                                       \
        }                                                                      \
    } while (0)

#define RUN_ALL_TESTS() ::testing::RunAll()

#endif  // GRAPHLILY_HIP_MINI_GTEST_H_
