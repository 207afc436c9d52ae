#pragma once
#include <memory>
#include <string>
namespace gr {
class logger {
public:
    explicit logger(const std::string &name);
    void debug(const std::string &msg);
    void info(const std::string &msg);
    void warn(const std::string &msg);
    void error(const std::string &msg);
};
typedef std::shared_ptr<logger> logger_ptr;
bool configure_default_loggers(gr::logger_ptr &l, gr::logger_ptr &d, const std::string &name);
}  // namespace gr
#define GR_LOG_DEBUG(log, msg) { log->debug(msg); }
#define GR_LOG_INFO(log, msg) { log->info(msg); }
#define GR_LOG_WARN(log, msg) { log->warn(msg); }
#define GR_LOG_ERROR(log, msg) { log->error(msg); }
