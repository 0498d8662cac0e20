// Bounce window, reference include/bounce_controller.h + src/bounce_controller.cpp:5-25.
// bounce 0 = directly visible emission, bounce k >= 1 = direct lighting at the k-th surface
// vertex; startBounce gates which terms are ADDED, lastBounce also ends the path.
#pragma once

#include <stdexcept>

namespace pathed {

class BounceController {
public:
    BounceController(int startBounce, int lastBounce)
        : m_startBounce(startBounce), m_lastBounce(lastBounce)
    {
        if (m_startBounce < 0 || !(m_lastBounce == -1 || m_startBounce <= m_lastBounce)) {
            throw std::runtime_error("BounceController: need 0 <= startBounce <= lastBounce (or lastBounce == -1)");
        }
    }

    bool checkCounts(int bounce) const
    {
        if (m_startBounce > bounce) { return false; }
        return !checkDone(bounce);
    }

    bool checkDone(int bounce) const
    {
        if (m_lastBounce == -1) { return false; }
        return bounce > m_lastBounce;
    }

    int startBounce() const { return m_startBounce; }
    int lastBounce() const { return m_lastBounce; }

private:
    int m_startBounce;
    int m_lastBounce;
};

}  // namespace pathed
