// tools/denseflow.cpp — the denseflow command line on the MI355X build.
//
// Keys, aliases, defaults, help text and exit codes follow /root/reference/tools/denseflow.cpp:8-47 and the
// `denseflow -h` listing in /root/reference/README.md:109-143 (cv::CommandLineParser semantics: `-k=v`,
// bare `-k` is a boolean presence, other tokens are positional).  One key is added:
// `-g, --gpus` (number of GPUs to shard a list.txt over, default 1).
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <cstdlib>
#include <map>

#include "dense_flow.h"
#include "utils.h"

namespace {

struct Key {
    vector<string> names; // as declared; printed sorted
    string def;           // "" = no default (flag-like: has() only when given)
    string help;
};

const vector<Key> kKeys = {
    {{"h", "help"}, "", "print help message"},
    {{"o", "outputDir"}, ".", "root dir of output"},
    {{"a", "algorithm"}, "tvl1", "optical flow algorithm (nv/tvl1/farn/brox)"},
    {{"s", "step"}, "0", "right - left (0 for img, non-0 for flow)"},
    {{"b", "bound"}, "32", "maximum of optical flow"},
    {{"nw", "newWidth"}, "0", "new width"},
    {{"nh", "newHeight"}, "0", "new height"},
    {{"ns", "newShort"}, "0", "short side length"},
    {{"cf", "classFolder"}, "", "outputDir/class/video/flow.jpg"},
    {{"if", "inputFrames"}, "", "inputs are frames"},
    {{"st", "saveType"}, "jpg", "save format type (png/h5/jpg)"},
    {{"f", "force"}, "", "regardless of the marked .done file"},
    {{"v", "verbose"}, "", "verbose"},
    {{"g", "gpus"}, "1", "number of GPUs to shard the input list over"},
};

class CommandLine {
  public:
    CommandLine(int argc, char **argv) {
        for (int i = 1; i < argc; ++i) {
            string tok = argv[i];
            if (tok.size() > 1 && tok[0] == '-') {
                tok = tok.substr(tok[1] == '-' ? 2 : 1);
                string val = "true";
                const size_t eq = tok.find('=');
                if (eq != string::npos) {
                    val = tok.substr(eq + 1);
                    tok = tok.substr(0, eq);
                }
                const Key *k = find(tok);
                if (!k) {
                    errors_.push_back("Unknown parameter " + tok); // reported by check()
                    continue;
                }
                values_[k->names[0]] = val;
            } else if (input_.empty()) {
                input_ = tok;
            }
        }
    }
    bool has(const string &name) const {
        const Key *k = find(name);
        return k && (values_.count(k->names[0]) || !k->def.empty());
    }
    string get(const string &name) const {
        const Key *k = find(name);
        if (!k)
            return "";
        auto it = values_.find(k->names[0]);
        return it != values_.end() ? it->second : k->def;
    }
    int getInt(const string &name) {
        const string v = get(name);
        try {
            size_t pos = 0;
            const int r = std::stoi(v, &pos);
            if (pos != v.size())
                throw std::invalid_argument(v);
            return r;
        } catch (...) {
            errors_.push_back("Parameter '" + name + "': can not convert: [" + v + "] to [int]");
            return 0;
        }
    }
    const string &input() const { return input_; }
    bool check() const { return errors_.empty(); }
    void printErrors() const {
        if (!errors_.empty())
            cout << endl << "ERRORS:" << endl;
        for (const string &e : errors_)
            cout << e << endl;
    }
    void printMessage() const {
        cout << "GPU optical flow extraction." << endl << "Usage: denseflow [params] input" << endl << endl;
        vector<const Key *> sorted;
        for (const Key &k : kKeys)
            sorted.push_back(&k);
        auto first = [](const Key *k) {
            vector<string> n = k->names;
            std::sort(n.begin(), n.end());
            return n[0];
        };
        std::sort(sorted.begin(), sorted.end(), [&](const Key *a, const Key *b) { return first(a) < first(b); });
        for (const Key *k : sorted) {
            vector<string> n = k->names;
            std::sort(n.begin(), n.end());
            cout << "\t";
            for (size_t i = 0; i < n.size(); ++i)
                cout << (i ? ", " : "") << (n[i].size() == 1 ? "-" : "--") << n[i];
            string def = k->def;
            if (k->names[0] == "h")
                def = "true"; // the parser shows the value `-h` was given
            if (!def.empty())
                cout << " (value:" << def << ")";
            cout << endl << "\t\t" << k->help << endl;
        }
        cout << endl << "\tinput" << endl << "\t\tfilename of video or folder of frames or a list.txt of those" << endl;
    }

  private:
    static const Key *find(const string &name) {
        for (const Key &k : kKeys)
            for (const string &n : k.names)
                if (n == name)
                    return &k;
        return nullptr;
    }
    std::map<string, string> values_;
    string input_;
    vector<string> errors_;
};

} // namespace

// dfx_device_count() in a forked child (exit status = count, capped at 255): the caller's process stays free of HIP.
static int device_count_in_a_child() {
    std::cout.flush();
    std::cerr.flush();
    const pid_t pid = fork();
    if (pid < 0)
        return 0;
    if (pid == 0)
        _exit(std::min(std::max(dfx_device_count(), 0), 255));
    int status = 0;
    if (waitpid(pid, &status, 0) != pid || !WIFEXITED(status))
        return 0;
    return WEXITSTATUS(status);
}

int main(int argc, char **argv) {
    try {
        CommandLine cmd(argc, argv);
        if (cmd.input().empty() || cmd.has("help")) {
            cmd.printMessage();
            return 0;
        }
        const path video_path(cmd.input());
        const path output_dir(cmd.get("outputDir"));
        const string algorithm = cmd.get("algorithm");
        const int step = cmd.getInt("step");
        const int bound = cmd.getInt("bound");
        const int new_width = cmd.getInt("newWidth");
        const int new_height = cmd.getInt("newHeight");
        const int new_short = cmd.getInt("newShort");
        const int gpus = cmd.getInt("gpus");
        const bool has_class = cmd.has("classFolder");
        const bool use_frames = cmd.has("inputFrames");
        const bool force = cmd.has("force");
        const string save_type = cmd.get("saveType");
        const bool verbose = cmd.has("verbose");
        if (!cmd.check()) {
            cmd.printErrors();
            return 0;
        }

        // reference: Mat::setDefaultAllocator(PAGE_LOCKED) (:49); DF_NO_PINNED=1 keeps pageable buffers
        Mat::setPageLocked(std::getenv("DF_NO_PINNED") == nullptr);

        vector<path> video_paths, output_dirs;
        bool is_record = false;
        if (video_path.extension() == ".txt") {
            is_record = true;
            std::ifstream ifs(video_path.string());
            string line;
            while (getline(ifs, line)) {
                if (line.empty())
                    continue;
                const path vidfile(line);
                const path cls = vidfile.parent_path().filename();
                const path outdir = has_class ? output_dir / cls / vidfile.stem() : output_dir / vidfile.stem();
                const path donedir = has_class ? output_dir / ".done" / cls : output_dir / ".done";
                if (!force && is_regular_file(donedir / vidfile.stem())) {
                    if (verbose)
                        cout << "skip " << cls / vidfile.stem() << endl;
                    continue;
                }
                create_directories(outdir);
                create_directories(donedir);
                video_paths.push_back(vidfile);
                output_dirs.push_back(outdir);
            }
        } else {
            const path outdir = output_dir / video_path.stem();
            create_directories(outdir);
            video_paths.push_back(video_path);
            output_dirs.push_back(outdir);
        }
        if (!video_paths.empty()) {
            vector<int> devices;
            // DF_PROCESSES=1 forks one process per pipeline: THIS process must not start the HIP runtime then (a runtime
            // inherited across fork() is undefined: the children hang or fail in dfx_create).  With DF_DEVICES the list is
            // taken as it stands (each child fails with the library's message on a bad index); without it the devices are
            // counted by a short-lived child of their own.
            const bool by_processes = std::getenv("DF_PROCESSES") != nullptr;
            if (std::getenv("STUB_TRACE_DEVICE_COUNT")) // tests/test_host_pipeline_stub.py: which process asks the library
                std::cerr << "stub: main pid " << (int)getpid() << std::endl;
            const bool forked = by_processes && std::getenv("DF_DEVICES");
            const int avail = forked ? (1 << 16) : std::max(by_processes ? device_count_in_a_child() : dfx_device_count(), 1);
            for (int g = 0; g < std::max(1, std::min(gpus, avail)); ++g)
                devices.push_back(g);
            if (const char *dl = std::getenv("DF_DEVICES")) { // explicit device list, e.g. "2,3" or (testing) "0,0"
                devices.clear();
                for (const char *q = dl; *q;) {
                    char *end = nullptr;
                    const long d = std::strtol(q, &end, 10);
                    if (end == q)
                        break;
                    if (d >= 0 && d < avail)
                        devices.push_back((int)d);
                    q = (*end == ',') ? end + 1 : end;
                }
                if (devices.empty())
                    devices.push_back(0);
            }
            calcDenseFlowVideoMultiGPU(video_paths, output_dirs, algorithm, step, bound, new_width, new_height,
                                       new_short, has_class, use_frames, save_type, is_record, verbose, devices);
        }
    } catch (const std::exception &ex) {
        cout << ex.what() << endl;
        return 1;
    }
    return 0;
}
